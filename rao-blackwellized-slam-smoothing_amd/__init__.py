"""MI355X-native Rao-Blackwellized particle filter / smoother (hot path of
manonkok/Rao-Blackwellized-SLAM-smoothing) -- host-side mirror of the reference interface over the
HIP C-ABI library.  The directory name contains hyphens, so import it with

    import importlib; rbpf = importlib.import_module("rao-blackwellized-slam-smoothing_amd")

(or `import rbpf_amd`, a one-line alias module at the repo root).
"""
from ._build import build, LIBPATH                                    # noqa: F401
from ._ffi import (RBPFError, load_library, EXPORTS,                  # noqa: F401
                   RBPF_OK, RBPF_ERR_INVALID_ARG, RBPF_ERR_UNSUPPORTED, RBPF_ERR_HIP, RBPF_ERR_NO_DEVICE,
                   RBPF_ERR_OUT_OF_MEMORY, RBPF_ERR_CHOL_FAILED, RBPF_ERR_STATE, RBPF_ERR_CALLBACK)
from .host import (particleFilter, particleSmoother, particleSmootherInformationForm,   # noqa: F401
                   DenseMagModel, DenseRadioModel, SparseVisualModel, dense_mag_prior, dense_radio_prior,
                   domain_cartesian_dx, eigenval, PhiloxRNG, ReplayRNG, FilterSession, sample, chol_weights, chol_sweep_probe, quat_helper,
                   GenericDenseModel, particle_filter_external, chol_refresh_in_use)


def device_count() -> int:
    """Visible HIP devices according to the library (0 without a GPU)."""
    return int(load_library().rbpf_device_count())
