"""nn_fac_amd -- MI355X (gfx950) inner-update engine behind nn_fac's HALS-NNLS / beta-MU hot path.

Sub-modules mirror the reference layout and signatures:
    nn_fac_amd.update_rules.nnls.hals_nnls_acc          (nn_fac/update_rules/nnls.py:24)
    nn_fac_amd.update_rules.mu.{mu_betadivmin, switch_alternate_mu, mu_tensorial}   (nn_fac/update_rules/mu.py:20,31,99)
    nn_fac_amd.utils.{errors, beta_divergence, initialize_factors}
    nn_fac_amd.nmf.{nmf, compute_nmf, one_nmf_step}     (nn_fac/nmf.py:19,196,332)
    nn_fac_amd.ntf.{ntf, compute_ntf, one_ntf_step}     (nn_fac/ntf.py:19,201,347)
    nn_fac_amd.ntd.{ntd, compute_ntd, one_ntd_step, one_ntd_step_mu}   (nn_fac/ntd.py:27,248,436,658)
Everything computes in libnnfac_hip.so (hand-written HIP, C ABI in include/nnfac_hip.h); there is no CPU fallback.
"""
__version__ = "0.1.0"
