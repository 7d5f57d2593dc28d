"""mendeliht.jl_amd -- MI355X-native IHT hot path behind MendelIHT.jl's API surface.

Host-side mirror of the reference interface for the hot path (`fit_iht`, `cv_iht`,
`iht`, `cross_validate`, `project_k!`, `project_group_sparse!`, `SnpLinAlg`) over the
C ABI of include/mendeliht_hip.h (libmendeliht_hip.so, hand-written HIP for gfx950).
The reference's own host language is Julia, which this image does not have; the
Julia glue a maintainer would add is julia/MendelIHTHip.jl.

There is no CPU fallback: every entry point raises if the HIP library or a GPU
is missing.
"""
from .api import (  # noqa: F401
    Bernoulli, IdentityLink, IHTResult, LogitLink, LogLink, MendelIHTError, MvNormal, NegativeBinomial,
    Normal, Poisson, SnpLinAlg, DenseMatrix, cross_validate, cv_iht, device_count, fit_iht, iht,
    library_path, mIHTResult, project_group_sparse, project_k, read_bed, standardize, lib, IHTSession,
    profile_enable, profile_read, profile_passes, cv_assignment, profile_counters, profile_exchange, EXCHANGE_KINDS, busy_union_ms, hash_folds, probe_set, using_probes,
    probes_library_path, iht_run_many_models, set_xtv_digits, set_step_mode, Gamma, InverseGaussian, ProbitLink,
    CloglogLink, CauchitLink, InverseLink, InverseSquareLink, SqrtLink, canonicallink, maf_weights, simulate_random_snparray, simulate_random_response, naive_impute,
)
from . import dist  # noqa: F401

__all__ = [
    "fit_iht", "cv_iht", "iht", "cross_validate", "project_k", "project_group_sparse", "SnpLinAlg",
    "DenseMatrix", "IHTResult", "mIHTResult", "Normal", "Bernoulli", "Poisson", "NegativeBinomial",
    "MvNormal", "Gamma", "InverseGaussian", "IdentityLink", "LogitLink", "LogLink", "ProbitLink", "CloglogLink",
    "CauchitLink", "InverseLink", "InverseSquareLink", "SqrtLink", "canonicallink", "maf_weights", "simulate_random_snparray", "simulate_random_response", "read_bed", "standardize", "device_count",
    "library_path", "MendelIHTError", "lib", "dist", "iht_run_many_models", "IHTSession", "naive_impute",
]
