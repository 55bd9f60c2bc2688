"""ctypes binding of libdeseq2_mi355x.so (include/deseq2_mi355x.h).

The shared library is built in-tree by `__graft_entry__.build()` (or
`make -C deseq2_amd/csrc`).  There is no fallback: if the library is missing or no
gfx950 device is usable, calls raise.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("DSQ_LIB") or os.path.join(_HERE, "libdeseq2_mi355x.so")   # DSQ_LIB: tuning builds only

DSQ_OK = 0
DSQ_ERR_ARG, DSQ_ERR_UNSUPPORTED, DSQ_ERR_DEVICE, DSQ_ERR_NOMEM, DSQ_ERR_VALUE = 1, 2, 3, 4, 5
DSQ_LAYOUT_R = 0
DSQ_LAYOUT_GENE_MAJOR = 1
DSQ_Y_INT32 = 0
DSQ_Y_FLOAT64 = 1
DSQ_MAX_P = 64

DSQ_ERR_FIT = 6
ERR_NAMES = {1: "DSQ_ERR_ARG", 2: "DSQ_ERR_UNSUPPORTED", 3: "DSQ_ERR_DEVICE", 4: "DSQ_ERR_NOMEM",
             5: "DSQ_ERR_VALUE", 6: "DSQ_ERR_FIT"}


class DsqError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (ERR_NAMES.get(code, "DSQ_ERR"), code, msg))
        self.code = code


class DsqFitBetaArgs(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("m", C.c_int32), ("p", C.c_int32), ("layout", C.c_int32), ("ld", C.c_int64),
        ("y", C.c_void_p), ("y_type", C.c_int32), ("x", C.c_void_p), ("nf", C.c_void_p),
        ("nf_is_vector", C.c_int32), ("alpha_hat", C.c_void_p), ("contrast", C.c_void_p),
        ("beta_mat", C.c_void_p), ("lambda_", C.c_void_p), ("weights", C.c_void_p),
        ("useWeights", C.c_int32), ("tol", C.c_double), ("maxit", C.c_int32), ("useQR", C.c_int32),
        ("minmu", C.c_double), ("cell_of", C.c_void_p), ("ncell", C.c_int32),
    ]


class DsqFitBetaOut(C.Structure):
    _fields_ = [
        ("beta_mat", C.c_void_p), ("beta_var_mat", C.c_void_p), ("iter", C.c_void_p),
        ("hat_diagonals", C.c_void_p), ("contrast_num", C.c_void_p), ("contrast_denom", C.c_void_p),
        ("deviance", C.c_void_p), ("mu", C.c_void_p), ("mu_floor", C.c_double),
    ]


class DsqFitDispArgs(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("m", C.c_int32), ("p", C.c_int32), ("layout", C.c_int32), ("ld", C.c_int64),
        ("y", C.c_void_p), ("y_type", C.c_int32), ("x", C.c_void_p), ("mu_hat", C.c_void_p),
        ("log_alpha", C.c_void_p), ("log_alpha_prior_mean", C.c_void_p),
        ("log_alpha_prior_sigmasq", C.c_double), ("min_log_alpha", C.c_double), ("kappa_0", C.c_double),
        ("tol", C.c_double), ("maxit", C.c_int32), ("usePrior", C.c_int32), ("weights", C.c_void_p),
        ("useWeights", C.c_int32), ("weightThreshold", C.c_double), ("useCR", C.c_int32),
        ("cell_of", C.c_void_p), ("ncell", C.c_int32),
    ]


class DsqFitDispOut(C.Structure):
    _fields_ = [
        ("log_alpha", C.c_void_p), ("iter", C.c_void_p), ("iter_accept", C.c_void_p),
        ("last_change", C.c_void_p), ("initial_lp", C.c_void_p), ("initial_dlp", C.c_void_p),
        ("last_lp", C.c_void_p), ("last_dlp", C.c_void_p), ("last_d2lp", C.c_void_p),
    ]


class DsqFitDispGridArgs(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("m", C.c_int32), ("p", C.c_int32), ("layout", C.c_int32), ("ld", C.c_int64),
        ("y", C.c_void_p), ("y_type", C.c_int32), ("x", C.c_void_p), ("mu_hat", C.c_void_p),
        ("disp_grid", C.c_void_p), ("ngrid", C.c_int32), ("log_alpha_prior_mean", C.c_void_p),
        ("log_alpha_prior_sigmasq", C.c_double), ("usePrior", C.c_int32), ("weights", C.c_void_p),
        ("useWeights", C.c_int32), ("weightThreshold", C.c_double), ("useCR", C.c_int32),
        ("cell_of", C.c_void_p), ("ncell", C.c_int32),
    ]


class DsqFitDispGridOut(C.Structure):
    _fields_ = [("log_alpha", C.c_void_p)]


class DsqPrefitArgs(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("m", C.c_int32), ("p", C.c_int32), ("layout", C.c_int32), ("ld", C.c_int64),
        ("y", C.c_void_p), ("y_type", C.c_int32), ("nf", C.c_void_p), ("nf_is_vector", C.c_int32),
        ("weights", C.c_void_p), ("useWeights", C.c_int32), ("q", C.c_void_p), ("a", C.c_void_p),
        ("r", C.c_void_p),
    ]


class DsqPrefitOut(C.Structure):
    _fields_ = [("baseMean", C.c_void_p), ("baseVar", C.c_void_p), ("allZero", C.c_void_p),
                ("roughDisp", C.c_void_p), ("beta_init", C.c_void_p)]


class DsqLogLikeArgs(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("m", C.c_int32), ("layout", C.c_int32), ("ld", C.c_int64), ("y", C.c_void_p),
        ("y_type", C.c_int32), ("mu", C.c_void_p), ("disp", C.c_void_p), ("weights", C.c_void_p),
        ("useWeights", C.c_int32),
    ]


class DsqInterceptArgs(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("m", C.c_int32), ("layout", C.c_int32), ("ld", C.c_int64), ("y", C.c_void_p),
        ("y_type", C.c_int32), ("nf", C.c_void_p), ("nf_is_vector", C.c_int32), ("weights", C.c_void_p),
        ("useWeights", C.c_int32), ("alpha", C.c_void_p), ("mu_floor", C.c_double),
    ]


class DsqInterceptOut(C.Structure):
    _fields_ = [("beta_log2", C.c_void_p), ("betaSE", C.c_void_p), ("mu", C.c_void_p), ("hat", C.c_void_p)]


class DsqOptimArgs(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("m", C.c_int32), ("p", C.c_int32), ("layout", C.c_int32), ("ld", C.c_int64),
        ("y", C.c_void_p), ("y_type", C.c_int32), ("x", C.c_void_p), ("nf", C.c_void_p), ("nf_is_vector", C.c_int32),
        ("alpha_hat", C.c_void_p), ("lambda_", C.c_void_p), ("weights", C.c_void_p), ("useWeights", C.c_int32),
        ("beta_start", C.c_void_p), ("minmu", C.c_double),
    ]


class DsqOptimOut(C.Structure):
    _fields_ = [("beta", C.c_void_p), ("betaSE", C.c_void_p), ("conv", C.c_void_p), ("mu", C.c_void_p),
                ("logLike", C.c_void_p)]


class DsqCooksArgs(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("m", C.c_int32), ("p", C.c_int32), ("layout", C.c_int32), ("ld", C.c_int64),
        ("y", C.c_void_p), ("y_type", C.c_int32), ("nf", C.c_void_p), ("nf_is_vector", C.c_int32),
        ("mu", C.c_void_p), ("H", C.c_void_p), ("cell_of", C.c_void_p), ("ncell", C.c_int32),
    ]


class DsqCooksOut(C.Structure):
    _fields_ = [("cooks", C.c_void_p), ("maxCooks", C.c_void_p), ("robustDisp", C.c_void_p)]


class DsqReplaceArgs(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("m", C.c_int32), ("layout", C.c_int32), ("ld", C.c_int64), ("y", C.c_void_p),
        ("y_type", C.c_int32), ("nf", C.c_void_p), ("nf_is_vector", C.c_int32), ("cooks", C.c_void_p),
        ("cooksCutoff", C.c_double), ("trim", C.c_double), ("replaceable", C.c_void_p),
    ]


class DsqReplaceOut(C.Structure):
    _fields_ = [("newCounts", C.c_void_p), ("replace", C.c_void_p)]


class DsqDeseqArgs(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("m", C.c_int32), ("p", C.c_int32), ("ld", C.c_int64), ("phases", C.c_int32),
        ("y", C.c_void_p), ("nf", C.c_void_p), ("nf_is_vector", C.c_int32), ("useWeights", C.c_int32),
        ("weights_raw", C.c_void_p), ("weights_norm", C.c_void_p), ("weights_floor", C.c_void_p),
        ("force_zero", C.c_void_p), ("x", C.c_void_p), ("q", C.c_void_p), ("a", C.c_void_p), ("r", C.c_void_p),
        ("xim", C.c_double), ("linearMu", C.c_int32),
        ("minDisp", C.c_double), ("kappa_0", C.c_double), ("dispTol", C.c_double), ("weightThreshold", C.c_double),
        ("outlierSD", C.c_double), ("betaTol", C.c_double), ("minmu", C.c_double),
        ("maxit", C.c_int32), ("useCR", C.c_int32), ("useQR", C.c_int32), ("betaMaxit", C.c_int32),
        ("disp_grid", C.c_void_p), ("ngrid", C.c_int32), ("expVarLogDisp", C.c_double),
        ("trend_mean", C.c_void_p), ("trend_disp", C.c_void_p), ("n_trend", C.c_int32),
        ("lambda_", C.c_void_p), ("min_log_alpha", C.c_double), ("workspace", C.c_void_p),
        ("workspace_bytes", C.c_int64), ("test", C.c_int32),
        ("cell_of", C.c_void_p), ("ncell", C.c_int32), ("replaceable", C.c_void_p),
        ("cooksCutoff", C.c_double), ("trim", C.c_double), ("do_replace", C.c_int32),
        ("x_red", C.c_void_p), ("q_red", C.c_void_p), ("a_red", C.c_void_p), ("r_red", C.c_void_p), ("p_red", C.c_int32),
        ("cell_of_red", C.c_void_p), ("ncell_red", C.c_int32), ("defer_finish", C.c_int32),
        ("n_refit_global", C.c_void_p), ("betaPrior", C.c_int32), ("x_prior", C.c_void_p), ("p_prior", C.c_int32),
        ("prior_expanded", C.c_int32), ("prior_intercept", C.c_int32), ("lambda_prior", C.c_void_p),
        ("fitType", C.c_int32), ("dispFit_in", C.c_void_p), ("trend_fit_in", C.c_void_p),
        ("dispPriorVar_in", C.c_double),
    ]


class DsqDeseqOut(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in (
        "baseMean", "baseVar", "allZero", "dispGeneEst", "dispGeneIter", "dispFit", "dispMAP", "dispersion",
        "dispIter", "dispOutlier", "beta", "betaSE", "stat", "pvalue", "betaConv", "betaIter", "logLike",
        "logLikeReduced", "maxCooks", "replace", "optim_geneest", "optim_test", "mu_hat", "mu", "H", "cooks",
        "replaceCounts", "status", "scalars", "mle_beta")]


class DsqDeseqHostArgs(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("m", C.c_int32), ("p", C.c_int32), ("counts", C.c_void_p), ("y_type", C.c_int32),
        ("x", C.c_void_p), ("sizeFactors", C.c_void_p), ("normalizationFactors", C.c_void_p), ("weights", C.c_void_p),
        ("q", C.c_void_p), ("r", C.c_void_p), ("xrinv", C.c_void_p),
        ("test", C.c_int32), ("x_reduced", C.c_void_p), ("q_reduced", C.c_void_p), ("r_reduced", C.c_void_p),
        ("p_reduced", C.c_int32), ("minReplicatesForReplace", C.c_double), ("cooksCutoff", C.c_double),
        ("expVarLogDisp", C.c_double), ("betaTol", C.c_double), ("minmu", C.c_double), ("maxit", C.c_int32),
        ("useQR", C.c_int32), ("disp_maxit", C.c_int32), ("useCR", C.c_int32), ("disp_grid", C.c_void_p),
        ("ngrid", C.c_int32),
        ("betaPrior", C.c_int32), ("x_prior", C.c_void_p), ("p_prior", C.c_int32), ("coef_factor", C.c_void_p),
        ("prior_coef_factor", C.c_void_p), ("prior_coef_src", C.c_void_p), ("betaPriorVar", C.c_void_p),
        ("fitType", C.c_int32), ("dispFit", C.c_void_p), ("geneEstOnly", C.c_int32),
        ("dispPriorVar", C.c_double),
    ]


class DsqBetaPriorArgs(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("p", C.c_int32), ("mle_beta", C.c_void_p), ("baseMean", C.c_void_p), ("dispFit", C.c_void_p),
        ("allZero", C.c_void_p), ("coef_factor", C.c_void_p), ("expanded", C.c_int32), ("p_prior", C.c_int32),
        ("prior_coef_factor", C.c_void_p), ("prior_coef_src", C.c_void_p), ("upperQuantile", C.c_double),
    ]


class DsqDeseqHostOut(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in (
        "baseMean", "baseVar", "allZero", "dispGeneEst", "dispGeneIter", "dispFit", "dispMAP", "dispersion",
        "dispIter", "dispOutlier", "beta", "betaSE", "stat", "pvalue", "betaConv", "betaIter", "logLike",
        "logLikeReduced", "maxCooks", "replace", "weightsFail", "mu", "H", "cooks", "replaceCounts")] + [
        ("dispersionFunction", C.c_double * 8), ("status", C.c_int32 * 16), ("betaPriorVar", C.c_double * DSQ_MAX_P),
        ("mle_beta", C.c_void_p)]


DSQ_PH_GENE_EST, DSQ_PH_TREND, DSQ_PH_MAP_TEST, DSQ_PH_OUTLIERS, DSQ_PH_FINISH, DSQ_PH_PRIOR = 1, 2, 4, 8, 16, 32
DSQ_PH_OUTLIERS_DETECT, DSQ_PH_OUTLIERS_REFIT = 64, 128
DSQ_ST = {k: i for i, k in enumerate((
    "N_NONZERO", "N_GRID_GENEEST", "N_TREND", "TREND_STATUS", "N_ABOVE_MIN", "N_GRID_MAP", "N_OPTIM_GENEEST",
    "N_OPTIM_TEST", "N_REPLACE", "N_REFIT", "N_GRID_GENEEST_REFIT", "N_GRID_MAP_REFIT", "N_OPTIM_GENEEST_REFIT",
    "N_OPTIM_TEST_REFIT"))}
DSQ_SC_FIT_USED = 4
DSQ_FIT = {"parametric": 0, "mean": 1, "parametric_or_mean": 2, "given": 3}
DSQ_ST_COUNT, DSQ_SC_COUNT = 16, 8

# every symbol include/deseq2_mi355x.h declares (tests check the .so exports all of them)
EXPORTED_SYMBOLS = [
    "dsq_fit_beta", "dsq_fit_beta_dev", "dsq_fit_disp", "dsq_fit_disp_dev", "dsq_fit_disp_grid",
    "dsq_fit_disp_grid_dev", "dsq_to_gene_major_f64", "dsq_to_gene_major_i32",
    "dsq_counts_f64_to_gene_major_i32", "dsq_from_gene_major_f64", "dsq_version", "dsq_last_error",
    "dsq_device_count", "dsq_set_device", "dsq_release_workspace", "dsq_test_math",
    "dsq_profile_enable", "dsq_profile_last_ms",
    "dsq_prefit_moments", "dsq_prefit_moments_dev", "dsq_nbinom_loglike", "dsq_nbinom_loglike_dev",
    "dsq_parametric_dispersion_fit", "dsq_parametric_dispersion_fit_dev",
    "dsq_fit_beta_rows", "dsq_fit_disp_rows", "dsq_fit_disp_grid_rows", "dsq_optim_rows",
    "dsq_intercept_fit", "dsq_intercept_fit_dev", "dsq_deseq_dev", "dsq_deseq_workspace_bytes",
    "dsq_profile_count", "dsq_profile_get",
    "dsq_deseq", "dsq_beta_prior_var", "dsq_weights_prep_dev", "dsq_xim_dev",
    "dsq_linear_mu", "dsq_linear_mu_dev", "dsq_cooks_distance", "dsq_cooks_distance_dev", "dsq_replace_outliers", "dsq_replace_outliers_dev",
]

_lib = None


def lib():
    """Load libdeseq2_mi355x.so (raises if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise FileNotFoundError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C deseq2_amd/csrc`.  deseq2_amd has no CPU fallback." % SO_PATH)
    # torch bundles its own HIP/HSA runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7).
    # A process must not end up with two runtimes, so when torch is installed it is imported
    # first: the engine library then binds to the runtime torch already loaded.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(SO_PATH)
    L.dsq_last_error.restype = C.c_char_p
    L.dsq_version.restype = C.c_int
    L.dsq_device_count.restype = C.c_int
    for name, a, o in (("dsq_fit_beta", DsqFitBetaArgs, DsqFitBetaOut),
                       ("dsq_fit_disp", DsqFitDispArgs, DsqFitDispOut),
                       ("dsq_fit_disp_grid", DsqFitDispGridArgs, DsqFitDispGridOut)):
        getattr(L, name).argtypes = [C.POINTER(a), C.POINTER(o)]
        getattr(L, name).restype = C.c_int
        getattr(L, name + "_dev").argtypes = [C.POINTER(a), C.POINTER(o), C.c_void_p]
        getattr(L, name + "_dev").restype = C.c_int
        getattr(L, name + "_rows").argtypes = [C.POINTER(a), C.POINTER(o), C.c_int64, C.c_int64]
        getattr(L, name + "_rows").restype = C.c_int
    L.dsq_to_gene_major_f64.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p]
    L.dsq_to_gene_major_i32.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p]
    L.dsq_counts_f64_to_gene_major_i32.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int64,
                                                    C.c_void_p, C.c_void_p]
    L.dsq_from_gene_major_f64.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p]
    L.dsq_test_math.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    L.dsq_prefit_moments.argtypes = [C.POINTER(DsqPrefitArgs), C.POINTER(DsqPrefitOut)]
    L.dsq_prefit_moments_dev.argtypes = [C.POINTER(DsqPrefitArgs), C.POINTER(DsqPrefitOut), C.c_void_p]
    L.dsq_nbinom_loglike.argtypes = [C.POINTER(DsqLogLikeArgs), C.c_void_p]
    L.dsq_nbinom_loglike_dev.argtypes = [C.POINTER(DsqLogLikeArgs), C.c_void_p, C.c_void_p]
    L.dsq_parametric_dispersion_fit.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    L.dsq_parametric_dispersion_fit_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                                     C.c_void_p]
    L.dsq_linear_mu.argtypes = [C.POINTER(DsqPrefitArgs), C.c_double, C.c_void_p]
    L.dsq_linear_mu_dev.argtypes = [C.POINTER(DsqPrefitArgs), C.c_double, C.c_void_p, C.c_void_p]
    L.dsq_intercept_fit.argtypes = [C.POINTER(DsqInterceptArgs), C.POINTER(DsqInterceptOut)]
    L.dsq_intercept_fit_dev.argtypes = [C.POINTER(DsqInterceptArgs), C.POINTER(DsqInterceptOut), C.c_void_p]
    L.dsq_optim_rows.argtypes = [C.POINTER(DsqOptimArgs), C.POINTER(DsqOptimOut)]
    L.dsq_cooks_distance.argtypes = [C.POINTER(DsqCooksArgs), C.POINTER(DsqCooksOut)]
    L.dsq_cooks_distance_dev.argtypes = [C.POINTER(DsqCooksArgs), C.POINTER(DsqCooksOut), C.c_void_p]
    L.dsq_replace_outliers.argtypes = [C.POINTER(DsqReplaceArgs), C.POINTER(DsqReplaceOut)]
    L.dsq_replace_outliers_dev.argtypes = [C.POINTER(DsqReplaceArgs), C.POINTER(DsqReplaceOut), C.c_void_p]
    L.dsq_deseq_dev.argtypes = [C.POINTER(DsqDeseqArgs), C.POINTER(DsqDeseqOut), C.c_void_p]
    L.dsq_weights_prep_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_double,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.dsq_xim_dev.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.dsq_deseq.argtypes = [C.POINTER(DsqDeseqHostArgs), C.POINTER(DsqDeseqHostOut)]
    L.dsq_beta_prior_var.argtypes = [C.POINTER(DsqBetaPriorArgs), C.c_void_p]
    L.dsq_deseq_workspace_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    L.dsq_deseq_workspace_bytes.restype = C.c_int64
    L.dsq_profile_get.argtypes = [C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_double)]
    L.dsq_set_device.argtypes = [C.c_int]
    L.dsq_profile_enable.argtypes = [C.c_int]
    L.dsq_profile_last_ms.restype = C.c_double
    _lib = L
    return L


def check(rc):
    if rc != DSQ_OK:
        raise DsqError(rc, lib().dsq_last_error().decode("utf-8", "replace"))
