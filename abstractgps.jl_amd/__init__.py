"""abstractgps.jl_amd — MI355X-native engine for the AbstractGPs.jl logpdf / posterior hot path.

Holds only what the path needs: `csrc/` (hand-written HIP kernels + the C ABI of include/gpmi355.h),
the Python mirror of the reference's FiniteGP API (`api.py`, what the tests run) and the Julia shim source
(`julia/HipGPs.jl`).  Multi-device fits are driven inside the library (`csrc/multi.hip`, gp_ctx_create_multi).

The directory name contains a dot, so import it through the root-level shim:
    import abstractgps_jl_amd as agp
"""
from .api import *  # noqa: F401,F403
from .api import (GP, ARDTransform, ColVecs, Context, FiniteGP, Kernel, Matern12Kernel, Matern32Kernel,
                  Matern52Kernel, PosteriorGP, RowVecs, ScaleTransform, SqExponentialKernel, cov, default_context,
                  kernelmatrix, loglikelihood, VFE, DTC, ExactInference, ApproxPosteriorGP, approx_log_evidence, elbo, inducing_points, logpdf, logpdf_and_grad, sqmahal, logdetcov, gradlogpdf, marginals, mean, mean_and_cov, mean_and_var, posterior, rand, rand_, update_posterior, var,
                  with_lengthscale)
from ._lib import GpmiError, PosDefException  # noqa: F401
