"""numpy restatement of Philox4x32-10 and of the normal/uniform draws the device HMC uses
(TEST ORACLE -- see oracle/__init__.py).

Philox is the counter-based generator of Salmon, Moraes, Dror & Shaw, "Parallel random
numbers: as easy as 1, 2, 3" (SC'11); the round constants and the known-answer vectors in
tests/test_oracle_pins.py are the published ones.  The reference itself draws its momenta
inside TFP (tfp.mcmc.HamiltonianMonteCarlo, bayes_vi/inference/mcmc/transition_kernels.py:122-128)
with TensorFlow's stateless generators, whose streams are not reproduced: parity with the
reference's sampler is statistical, parity between this oracle and the HIP kernels is exact
for the integer stream and to float rounding for the transformed draws.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(counter, key):
    """counter[..., 4], key[..., 2] uint32 -> [..., 4] uint32."""
    c = np.asarray(counter, dtype=np.uint64).copy()
    k = np.broadcast_to(np.asarray(key, dtype=np.uint64), c.shape[:-1] + (2,)).copy()
    for _ in range(10):
        p0 = M0 * c[..., 0]
        p1 = M1 * c[..., 2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK
        c = np.stack([(hi1 ^ c[..., 1] ^ k[..., 0]) & MASK, lo1, (hi0 ^ c[..., 3] ^ k[..., 1]) & MASK, lo0], -1)
        k = np.stack([(k[..., 0] + np.uint64(W0)) & MASK, (k[..., 1] + np.uint64(W1)) & MASK], -1)
    return c.astype(np.uint32)


def draw_blocks(seed, step, stream_id, n_chains, n_blocks):
    """Counter layout of the HIP kernels: (step lo, step hi, chain, block | stream << 28), key = seed halves."""
    chain = np.arange(n_chains, dtype=np.uint64)[:, None] + np.zeros((1, n_blocks), np.uint64)
    blk = np.arange(n_blocks, dtype=np.uint64)[None, :] + np.zeros((n_chains, 1), np.uint64)
    ctr = np.stack([np.full_like(chain, step & 0xFFFFFFFF), np.full_like(chain, (step >> 32) & 0xFFFFFFFF),
                    chain, blk | np.uint64(stream_id << 28)], -1)
    return philox4x32_10(ctr, np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], np.uint64))


def u01(x):
    return ((x >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)


def normals(seed, step, n_chains, n_params):
    """The momenta of bvi_hmc_begin: Box-Muller pairs on 24-bit uniforms, float32."""
    nb = (n_params + 3) // 4
    r = draw_blocks(seed, step, 0, n_chains, nb)
    u = u01(r).astype(np.float64)
    r0 = np.sqrt(-2.0 * np.log(u[..., 0])); t0 = 2 * np.pi * u[..., 1]
    r1 = np.sqrt(-2.0 * np.log(u[..., 2])); t1 = 2 * np.pi * u[..., 3]
    out = np.stack([r0 * np.cos(t0), r0 * np.sin(t0), r1 * np.cos(t1), r1 * np.sin(t1)], -1)
    return out.reshape(n_chains, nb * 4)[:, :n_params].astype(np.float32)


def log_uniform(seed, step, n_chains):
    """The accept uniform of bvi_hmc_end: stream 1, block 0, first word."""
    r = draw_blocks(seed, step, 1, n_chains, 1)
    return np.log(u01(r[:, 0, 0]).astype(np.float64))
