"""The parameter objects of the multilevel ILU++ preconditioner, as the reference binds them (src/binding.cpp:462-592):
``iluplusplus_precond_parameter`` with its attributes, ``default_configuration`` and the ``PREPROCESSING`` sequence.

The values are those of iluplusplus_precond_parameter::default_parameters (parameters_implementation.h:430-501), the presets those of
default_configuration (:538-830) and init (:832-1700) -- restated here as data for the presets this package documents; a preset number
outside that list raises NotImplementedError rather than guessing.  What the GPU engine has built of the family is decided when a
preconditioner is constructed (``_to_ml_params``): a parameter set it cannot honour is refused there, never silently changed.
"""
from . import _native

# preprocessing_type names (functions_implementation.h:94-127) the engine has kernels for -> ILUPP_PRE_* of include/ilupp_hip.h
_BUILT_STEPS = {"NORMALIZE_COLUMNS": 1, "NORMALIZE_ROWS": 2, "PQ_ORDERING": 3, "MAX_WEIGHTED_MATCHING_ORDERING": 4,
                "DD_SYMM_MOVE_CORNER_ORDERING_IM": 5, "UNIT_OR_ZERO_DIAGONAL_SCALING": 6, "SPARSE_FIRST_ORDERING": 7, "SYMM_PQ": 8}


class preprocessing_sequence(list):
    """the steps run on the matrix of every level before it is factored (orderings.h:58-140); a list of step names"""

    def _set(self, *names):
        self[:] = list(names)

    def set_none(self):
        self._set()

    def set_normalize(self):
        self._set("NORMALIZE_COLUMNS", "NORMALIZE_ROWS")

    def set_PQ(self):
        self._set("NORMALIZE_COLUMNS", "NORMALIZE_ROWS", "PQ_ORDERING")

    def set_MAX_WEIGHTED_MATCHING_ORDERING(self):
        self._set("MAX_WEIGHTED_MATCHING_ORDERING")

    def set_NORM_MAX_WEIGHTED_MATCHING_ORDERING(self):
        self._set("NORMALIZE_COLUMNS", "NORMALIZE_ROWS", "MAX_WEIGHTED_MATCHING_ORDERING")

    def set_MAX_WEIGHTED_MATCHING_ORDERING_PQ(self):
        self._set("MAX_WEIGHTED_MATCHING_ORDERING", "PQ_ORDERING")

    def set_MAX_WEIGHTED_MATCHING_ORDERING_UNIT_DIAG(self):
        self._set("MAX_WEIGHTED_MATCHING_ORDERING", "UNIT_OR_ZERO_DIAGONAL_SCALING")

    def set_MAX_WEIGHTED_MATCHING_ORDERING_DD_MOV_COR_IM(self):
        self._set("MAX_WEIGHTED_MATCHING_ORDERING", "DD_SYMM_MOVE_CORNER_ORDERING_IM")

    def set_NORM_MAX_WEIGHTED_MATCHING_ORDERING_DD_MOV_COR_IM(self):
        self._set("NORMALIZE_COLUMNS", "NORMALIZE_ROWS", "MAX_WEIGHTED_MATCHING_ORDERING", "DD_SYMM_MOVE_CORNER_ORDERING_IM")

    def set_MAX_WEIGHTED_MATCHING_ORDERING_SYM_PQ(self):
        self._set("MAX_WEIGHTED_MATCHING_ORDERING", "SYMM_PQ")

    def set_NORM_MAX_WEIGHTED_MATCHING_ORDERING_SYM_PQ(self):
        self._set("NORMALIZE_COLUMNS", "NORMALIZE_ROWS", "MAX_WEIGHTED_MATCHING_ORDERING", "SYMM_PQ")

    def set_SPARSE_FIRST(self):
        self._set("SPARSE_FIRST_ORDERING")

    def set_SPARSE_FIRST_MAX_WEIGHTED_MATCHING_ORDERING(self):
        self._set("SPARSE_FIRST_ORDERING", "MAX_WEIGHTED_MATCHING_ORDERING")

    def set_SPARSE_FIRST_MAX_WEIGHTED_MATCHING_ORDERING_DD_MOV_COR_IM(self):
        self._set("SPARSE_FIRST_ORDERING", "MAX_WEIGHTED_MATCHING_ORDERING", "DD_SYMM_MOVE_CORNER_ORDERING_IM")

    def to_names(self):
        return list(self)


# default_parameters(), parameters_implementation.h:430-501
_DEFAULTS = dict(
    fill_in=10000, threshold=0.0, piv_tol=1.0, GLOBAL_COMMENT="default parameters", PRECON_PARAMETER=0, PQ_ALGORITHM=0, PQ_THRESHOLD=0.0,
    MAX_LEVELS=100, MEMORY_MAX_LEVELS=100, MAX_FILLIN_IS_INF=True, BEGIN_TOTAL_PIV=True, TOTAL_PIV=1, MIN_ML_SIZE=0,
    USE_FINAL_THRESHOLD=False, FINAL_THRESHOLD=0.0, VARY_THRESHOLD_FACTOR=1.0, THRESHOLD_SHIFT_SCHUR=0.0, PERMUTE_ROWS=3,
    EXTERNAL_FINAL_ROW=False, MIN_ELIM_FACTOR=0.5, REQUIRE_ZERO_SCHUR=False, REQ_ZERO_SCHUR_SIZE=0, EXT_MIN_ELIM_FACTOR=0.0,
    FINAL_ROW_CRIT=-1, SMALL_PIVOT_TERMINATES=False, MIN_PIVOT=1e-2, USE_THRES_ZERO_SCHUR=False, THRESHOLD_ZERO_SCHUR=1e-6,
    MIN_SIZE_ZERO_SCHUR=100, ROW_U_MAX=1.5, MOVE_LEVEL_FACTOR=2.0, MOVE_LEVEL_THRESHOLD=10.0, USE_MAX_AS_MOVE=True, MEM_FACTOR=3.0,
    VARIABLE_MEM=0, USE_STANDARD_DROPPING=False, USE_STANDARD_DROPPING2=False, USE_INVERSE_DROPPING=False, USE_WEIGHTED_DROPPING=False,
    USE_WEIGHTED_DROPPING2=False, USE_ERR_PROP_DROPPING=True, USE_ERR_PROP_DROPPING2=False, USE_PIVOT_DROPPING=False,
    INIT_WEIGHTS_LU=1.0, DROP_TYPE_L=0, DROP_TYPE_U=0, BANDWIDTH_MULTIPLIER=0.5, BANDWIDTH_OFFSET=0, SIZE_TABLE_POS_WEIGHTS=100,
    WEIGHT_TABLE_TYPE=1, SCALE_WEIGHT_INVDIAG=False, SCALE_WGT_MAXINVDIAG=False, WEIGHT_STANDARD_DROP=1.0, WEIGHT_STANDARD_DROP2=1.0,
    WEIGHT_INVERSE_DROP=1.0, WEIGHT_WEIGHTED_DROP=1.0, WEIGHT_ERR_PROP_DROP=1.0, WEIGHT_ERR_PROP_DROP2=1.0, WEIGHT_PIVOT_DROP=1.0,
    COMBINE_FACTOR=0, NEUTRAL_ELEMENT=0.0, MIN_WEIGHT=1.0, WEIGHTED_DROPPING=True, SUM_DROPPING=False, USE_POS_COMPRESS=False,
    POST_FACT_THRESHOLD=0.0, SCHUR_COMPLEMENT=0)

# init(), parameters_implementation.h:872-934 and :1533-1601: what a precon_parameter changes on top of the defaults
_NO_PIVOT = dict(PERMUTE_ROWS=0, TOTAL_PIV=0, piv_tol=0.0, SMALL_PIVOT_TERMINATES=True, MIN_ELIM_FACTOR=0.0)
_LESS_MEMORY = dict(THRESHOLD_SHIFT_SCHUR=1e-3, MAX_FILLIN_IS_INF=False, fill_in=500)
_PRECON = {
    0: {},
    1: dict(USE_INVERSE_DROPPING=True, USE_ERR_PROP_DROPPING=False),
    2: dict(USE_WEIGHTED_DROPPING=True, USE_ERR_PROP_DROPPING=False),
    3: dict(USE_STANDARD_DROPPING=True, USE_ERR_PROP_DROPPING=False),
    5: dict(SCHUR_COMPLEMENT=1),
    10: dict(_NO_PIVOT),
    11: dict(_NO_PIVOT, USE_INVERSE_DROPPING=True, USE_ERR_PROP_DROPPING=False),
    12: dict(_NO_PIVOT, USE_WEIGHTED_DROPPING=True, USE_ERR_PROP_DROPPING=False),
    13: dict(_NO_PIVOT, USE_STANDARD_DROPPING=True, USE_ERR_PROP_DROPPING=False),
    15: dict(_NO_PIVOT, SCHUR_COMPLEMENT=1),
    1000: dict(_LESS_MEMORY),
    1010: dict(_NO_PIVOT, **_LESS_MEMORY),
}
# default_configuration(), :538-609: (preprocessing, precon_parameter)
_CONFIG = {
    0: ("set_PQ", 0), 1: ("set_PQ", 10),
    10: ("set_MAX_WEIGHTED_MATCHING_ORDERING", 0), 11: ("set_MAX_WEIGHTED_MATCHING_ORDERING_DD_MOV_COR_IM", 10),
    12: ("set_SPARSE_FIRST_MAX_WEIGHTED_MATCHING_ORDERING", 10), 13: ("set_SPARSE_FIRST_MAX_WEIGHTED_MATCHING_ORDERING_DD_MOV_COR_IM", 10),
    1000: ("set_PQ", 1000), 1001: ("set_PQ", 1010),
    1010: ("set_MAX_WEIGHTED_MATCHING_ORDERING", 1000), 1011: ("set_MAX_WEIGHTED_MATCHING_ORDERING_DD_MOV_COR_IM", 1010),
    1012: ("set_SPARSE_FIRST_MAX_WEIGHTED_MATCHING_ORDERING", 1010), 1013: ("set_SPARSE_FIRST_MAX_WEIGHTED_MATCHING_ORDERING_DD_MOV_COR_IM", 1010),
}


class iluplusplus_precond_parameter:
    """parameters of the multilevel ILU++ preconditioner (reference: parameters.h:120-330, bound at binding.cpp:462-545)"""

    def __init__(self):
        self._defaults()

    def _defaults(self):
        for k, v in _DEFAULTS.items():
            setattr(self, k, v)
        self.PREPROCESSING = preprocessing_sequence()
        self.PREPROCESSING.set_PQ()

    def init(self, preprocessing, precon_parameter=0, global_comment=""):
        """iluplusplus_precond_parameter::init (:832-836 + the switch)"""
        if precon_parameter not in _PRECON:
            raise NotImplementedError("precon_parameter %d is not among the presets this package restates (%s)"
                                      % (precon_parameter, sorted(_PRECON)))
        self._defaults()
        self.PRECON_PARAMETER = precon_parameter
        self.GLOBAL_COMMENT = global_comment
        self.PREPROCESSING = preprocessing_sequence(preprocessing)
        for k, v in _PRECON[precon_parameter].items():
            setattr(self, k, v)

    def default_configuration(self, configuration=0):
        if configuration not in _CONFIG:
            raise NotImplementedError("default_configuration(%d) is not among the presets this package restates (%s)"
                                      % (configuration, sorted(_CONFIG)))
        how, precon = _CONFIG[configuration]
        seq = preprocessing_sequence()
        getattr(seq, how)()
        self.init(seq, precon, "")

    def set(self, fill_in, threshold, piv_tol):
        self.fill_in, self.threshold, self.piv_tol = fill_in, threshold, piv_tol

    # the use_only_* members (parameters_implementation.h:360-428)
    def _only(self, name):
        for k in ("USE_STANDARD_DROPPING", "USE_STANDARD_DROPPING2", "USE_INVERSE_DROPPING", "USE_WEIGHTED_DROPPING", "USE_WEIGHTED_DROPPING2",
                  "USE_ERR_PROP_DROPPING", "USE_ERR_PROP_DROPPING2", "USE_PIVOT_DROPPING"):
            setattr(self, k, k == name)

    def use_only_standard_dropping1(self):
        self._only("USE_STANDARD_DROPPING")

    def use_only_standard_dropping2(self):
        self._only("USE_STANDARD_DROPPING2")

    def use_only_inverse_dropping(self):
        self._only("USE_INVERSE_DROPPING")

    def use_only_weighted_dropping1(self):
        self._only("USE_WEIGHTED_DROPPING")

    def use_only_weighted_dropping2(self):
        self._only("USE_WEIGHTED_DROPPING2")

    def use_only_error_propagation_dropping1(self):
        self._only("USE_ERR_PROP_DROPPING")

    def use_only_error_propagation_dropping2(self):
        self._only("USE_ERR_PROP_DROPPING2")

    def use_only_pivot_dropping(self):
        self._only("USE_PIVOT_DROPPING")

    # ---- what the engine can do with this parameter set -------------------------------------------------------------------------
    def _uses_partial_iluc(self):
        """the test of make_preprocessed_multilevelILUCDP (preconditioner_implementation.h:1385-1390)"""
        return ((self.PERMUTE_ROWS == 0 or (self.PERMUTE_ROWS == 1 and not self.EXTERNAL_FINAL_ROW))
                and (not self.BEGIN_TOTAL_PIV or self.TOTAL_PIV == 0) and self.piv_tol == 0.0)

    def _to_ml_params(self):
        """the C-ABI parameter block, or NotImplementedError naming what this build lacks"""
        def refuse(what):
            raise NotImplementedError("ilupp_amd: the multilevel ILU++ preconditioner is built with the dropping rules that need no estimates "
                                      "carried over the steps (standard, error propagation, pivot) and the plain Schur complement; "
                                      + what + " is not built")
        if self.PRECON_PARAMETER < 0:
            refuse("PRECON_PARAMETER < 0 (reserved for external solvers)")
        if self.PERMUTE_ROWS not in (0, 1, 2, 3) or self.TOTAL_PIV not in (0, 1, 2):
            raise ValueError("choose permissible value for PERMUTE_ROWS / TOTAL_PIV!")
        if not self._uses_partial_iluc() and not (-1 <= self.FINAL_ROW_CRIT <= 9):
            refuse("FINAL_ROW_CRIT = %r with the pivoting factorisation (rows ordered by weights instead of counts)" % (self.FINAL_ROW_CRIT,))
        # (the inverse-based and weighted rules accumulate estimates over the steps in their sequential order: with them both factorisations
        #  run as chains walked by one wave, pilucdp.hip)
        rules = 0
        for bit, k in ((1, "USE_STANDARD_DROPPING"), (2, "USE_STANDARD_DROPPING2"), (4, "USE_ERR_PROP_DROPPING"), (8, "USE_ERR_PROP_DROPPING2"),
                       (16, "USE_PIVOT_DROPPING"), (32, "USE_INVERSE_DROPPING"), (64, "USE_WEIGHTED_DROPPING"), (128, "USE_WEIGHTED_DROPPING2")):
            if getattr(self, k):
                rules |= bit
        checks = [("DROP_TYPE_L", 0), ("DROP_TYPE_U", 0), ("SCHUR_COMPLEMENT", 0), ("EXTERNAL_FINAL_ROW", False),
                  ("REQUIRE_ZERO_SCHUR", False), ("USE_THRES_ZERO_SCHUR", False), ("WEIGHTED_DROPPING", True), ("SUM_DROPPING", False),
                  ("SCALE_WGT_MAXINVDIAG", False), ("USE_POS_COMPRESS", False)]
        for name, want in checks:
            if getattr(self, name) != want:
                refuse("%s = %r" % (name, getattr(self, name)))
        if self.FINAL_ROW_CRIT >= 11:
            refuse("FINAL_ROW_CRIT = %r" % (self.FINAL_ROW_CRIT,))
        if self.PQ_ALGORITHM not in (0, 2) and "PQ_ORDERING" in self.PREPROCESSING:
            refuse("PQ_ALGORITHM = %r" % (self.PQ_ALGORITHM,))
        steps = []
        for s in self.PREPROCESSING:
            if s not in _BUILT_STEPS:
                refuse("the preprocessing step " + str(s))
            steps.append(_BUILT_STEPS[s])
        if len(steps) > 8:
            refuse("more than 8 preprocessing steps")
        p = _native.MLParams()
        p.threshold = float(self.threshold)
        p.n_preprocessing = len(steps)
        for i, s in enumerate(steps):
            p.preprocessing[i] = s
        p.pq_threshold = float(self.PQ_THRESHOLD)
        p.max_levels = int(self.MAX_LEVELS)
        p.min_ml_size = int(self.MIN_ML_SIZE)
        p.small_pivot_terminates = 1 if self.SMALL_PIVOT_TERMINATES else 0
        p.min_pivot = float(self.MIN_PIVOT)
        p.min_elim_factor = float(self.MIN_ELIM_FACTOR)
        p.threshold_shift_schur = float(self.THRESHOLD_SHIFT_SCHUR)
        p.vary_threshold_factor = float(self.VARY_THRESHOLD_FACTOR)
        p.use_final_threshold = 1 if self.USE_FINAL_THRESHOLD else 0
        p.final_threshold = float(self.FINAL_THRESHOLD)
        p.max_fill_in = 0 if self.MAX_FILLIN_IS_INF else max(1, int(self.fill_in))      # partialILUC :1440-1447 (clamped to the level's size there)
        p.drop_rules = rules
        p.weight_standard_drop, p.weight_standard_drop2 = float(self.WEIGHT_STANDARD_DROP), float(self.WEIGHT_STANDARD_DROP2)
        p.weight_err_prop_drop, p.weight_err_prop_drop2 = float(self.WEIGHT_ERR_PROP_DROP), float(self.WEIGHT_ERR_PROP_DROP2)
        p.weight_pivot_drop = float(self.WEIGHT_PIVOT_DROP)
        p.weight_inverse_drop = float(self.WEIGHT_INVERSE_DROP)
        p.weight_weighted_drop, p.init_weights_lu = float(self.WEIGHT_WEIGHTED_DROP), float(self.INIT_WEIGHTS_LU)
        p.combine_factor = int(self.COMBINE_FACTOR) if int(self.COMBINE_FACTOR) in (0, 1, 2, 3) else 0       # combine(): default branch = max
        p.neutral_element, p.min_weight = float(self.NEUTRAL_ELEMENT), float(self.MIN_WEIGHT)
        p.scale_weight_invdiag = 1 if self.SCALE_WEIGHT_INVDIAG else 0
        p.piv_tol = float(self.piv_tol)
        p.permute_rows, p.total_piv, p.begin_total_piv = int(self.PERMUTE_ROWS), int(self.TOTAL_PIV), 1 if self.BEGIN_TOTAL_PIV else 0
        p.final_row_crit = int(self.FINAL_ROW_CRIT)
        p.move_level_factor, p.row_u_max = float(self.MOVE_LEVEL_FACTOR), float(self.ROW_U_MAX)
        return p
