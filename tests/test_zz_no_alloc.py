"""GPU: the allocation-free entry point bvi_normal_iid_logp_grad_noalloc (scratch in a device array inside the
library) against bvi_normal_iid_logp_grad (caller workspace) through the C ABI: two calls in a row, the second on a
smaller input, so that stale partial sums of the first call would show; and the size limit."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_noalloc_entry_point_matches_workspace_entry_point(gpu_device):
    from bayes_vi_amd import _native as nat
    lib = nat.lib()
    ws = nat.Workspace(torch.device(gpu_device))
    g = torch.Generator(device=gpu_device); g.manual_seed(3)
    for n, c in ((3_000_001, 200), (70_001, 7), (1_000, 256), (5, 1)):
        y = torch.empty(n, device=gpu_device).normal_(7.0, 3.0, generator=g)
        th = torch.stack([7 + 0.1 * torch.randn(c, generator=g, device=gpu_device),
                          3 + 0.2 * torch.rand(c, generator=g, device=gpu_device)], -1).contiguous()
        ref_lp, ref_g = nat.normal_iid_logp_grad(y, th, ws)
        lp = torch.full((c,), float('nan'), device=gpu_device)
        gr = torch.full((c, 2), float('nan'), device=gpu_device)
        before = torch.cuda.memory_stats()['allocation.all.allocated']      # a counter: frees elsewhere do not move it
        st = lib.bvi_normal_iid_logp_grad_noalloc(nat._ptr(y), n, nat._ptr(th), c, nat._ptr(lp), nat._ptr(gr),
                                                  nat._stream())
        assert st == 0, lib.bvi_status_string(st)
        assert torch.cuda.memory_stats()['allocation.all.allocated'] == before
        torch.cuda.synchronize()
        assert torch.equal(lp, ref_lp) and torch.equal(gr, ref_g)
        # value only
        lp2 = torch.empty(c, device=gpu_device)
        st = lib.bvi_normal_iid_logp_grad_noalloc(nat._ptr(y), n, nat._ptr(th), c, nat._ptr(lp2), None, nat._stream())
        assert st == 0 and torch.equal(lp2, ref_lp)
    # The counter above only sees torch's caching allocator.  Two checks that see the library's own calls: the
    # device's free memory (hipMemGetInfo) is the same before and after a call, and a call recorded into a HIP graph
    # -- where hipMalloc / hipFree are not permitted on the capturing stream -- captures and replays to the same result.
    torch.cuda.synchronize()
    free_before, _ = torch.cuda.mem_get_info()
    st = lib.bvi_normal_iid_logp_grad_noalloc(nat._ptr(y), n, nat._ptr(th), c, nat._ptr(lp), nat._ptr(gr), nat._stream())
    torch.cuda.synchronize()
    assert st == 0 and torch.cuda.mem_get_info()[0] == free_before
    lp.fill_(float('nan')); gr.fill_(float('nan'))
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        st = lib.bvi_normal_iid_logp_grad_noalloc(nat._ptr(y), n, nat._ptr(th), c, nat._ptr(lp), nat._ptr(gr),
                                                  nat._stream())
    assert st == 0
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(lp, ref_lp) and torch.equal(gr, ref_g)
    # beyond the fixed scratch: a clear error, nothing launched
    c = nat.NOALLOC_MAX_CHAINS + 1
    th = torch.ones(c, 2, device=gpu_device)
    lp = torch.zeros(c, device=gpu_device)
    st = lib.bvi_normal_iid_logp_grad_noalloc(nat._ptr(y), y.shape[0], nat._ptr(th), c, nat._ptr(lp), None,
                                              nat._stream())
    assert st == nat.ERR_TOO_LARGE
    assert b'allocation-free' in lib.bvi_status_string(st)
    assert float(lp.abs().sum()) == 0.0
