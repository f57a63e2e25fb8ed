"""MI355X-native hot path of bayes_vi: batched log-posterior value and gradient.

Host-side mirror of the reference's `Model` / `MCMC.fit` / `VI.fit` surface over
hand-written gfx950 HIP kernels reached through the C ABI of libbvi_hip.so.
"""
