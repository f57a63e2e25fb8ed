"""CPU oracle for the bayes_vi log-posterior hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the shipped product: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker.  The product path
(``bayesian-inference_amd/``) never imports this package and fails loudly when
the HIP library is missing.

Parity status: the reference (bayes_vi 0.3.0) ships no tests or fixtures and
its arithmetic lives in tensorflow-probability 0.12.1 (requirements.txt:108),
which is not installed here, so the reference's own outputs could not be
generated: *parity against TFP output is unpinned*.  The oracle is instead
pinned by (i) the closed-form densities cross-checked against scipy.stats and
torch.distributions, (ii) fp64 central finite differences for every gradient,
(iii) the exact harmonic-oscillator leapfrog orbit shown in the reference's
README.md:571-591, and (iv) the sufficient-statistic posterior moments in
univariate_gaussian_experiment/tables (see tests/test_oracle_pins.py).

Modules
-------
closed_form     fp64 numpy closed forms: priors, likelihood families, bijectors.
restated_torch  fp32 torch-CPU restatement of how TFP evaluates the same maths
                (materialised [chains, rows] tensors + autograd).  Timed as the
                CPU baseline.
leapfrog        the leapfrog recipe of bayes_vi/utils/symplectic_integrators.py
                and an HMC step with injected momenta / uniforms.
"""
