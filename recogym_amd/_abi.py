"""ctypes mirror of include/recogym_hip.h and the loader of librecogym_hip.so.

There is no CPU fallback: if the HIP library is missing or no HIP device is visible the
compute entry points raise.  (The float64 CPU restatement under oracle/ is test
infrastructure and is never imported from this package.)
"""
import ctypes as C
import os

RG_ABI_VERSION = 6

RG_STATE_ORGANIC, RG_STATE_BANDIT, RG_STATE_STOP = 0, 1, 2

RG_POLICY_UNIFORM_ENV = 0
RG_POLICY_RANDOM_AGENT = 1
RG_POLICY_ORGANIC_USER_COUNT = 2
RG_POLICY_EXTERNAL = 3
RG_POLICY_LAST_VIEW_TABLE = 4
RG_POLICY_LOGREG_FROZEN = 5

RG_EV_BANDIT = 0x80000000
RG_EV_CLICK = 0x40000000
RG_EV_PHANTOM = 0x20000000
RG_EV_INDEX_MASK = 0x1FFFFFFF

(RG_CNT_ORGANIC, RG_CNT_BANDIT, RG_CNT_CLICKS, RG_CNT_PHANTOM, RG_CNT_LIVE, RG_CNT_STEP,
 RG_CNT_LOG_ROWS, RG_CNT_LOG_DROPPED, RG_CNT_EXACT_DRAWS, RG_CNT_HIST_OVERFLOW,
 RG_CNT_EXACT_SWEEPS, RG_CNT_EXACT_OVERFLOW, RG_CNT_LR_ACTS, RG_CNT_LR_ROWS, RG_CNT_LR_EXACT, RG_CNT_MEMO_HITS) = range(16)
RG_CNT_ANCHORED = 24
RG_CNT_BAD_ACTION = 25
RG_CNT_N = 32

RG_ERRORS = {-1: 'RG_EINVAL', -2: 'RG_ENODEV', -3: 'RG_ENOMEM', -4: 'RG_ESTATE', -5: 'RG_ELIMIT'}


class RgConfig(C.Structure):
    """struct rg_config (include/recogym_hip.h)."""
    _fields_ = [
        ('num_products', C.c_uint32),
        ('K', C.c_uint32),
        ('seed', C.c_uint64),
        ('policy_seed', C.c_uint64),
        ('trans_cdf', (C.c_double * 3) * 2),
        ('sigma_omega_initial', C.c_double),
        ('sigma_omega', C.c_double),
        ('change_omega_for_bandits', C.c_uint32),
        ('policy', C.c_uint32),
        ('ouc_select_randomly', C.c_uint32),
        ('ouc_exploit_explore', C.c_uint32),
        ('ouc_reverse_pop', C.c_uint32),
        ('ouc_history_cap', C.c_uint32),
        ('ouc_epsilon', C.c_double),
        ('time_mode', C.c_uint32),
        ('env_kind', C.c_uint32),
        ('time_mu', C.c_double),
        ('time_sigma', C.c_double),
        ('lr_select_randomly', C.c_uint32),
        ('reserved1', C.c_uint32),
    ]


class RgEvent(C.Structure):
    """struct rg_event: one 16-byte device log row."""
    _fields_ = [('u', C.c_uint32), ('t', C.c_uint32), ('code', C.c_uint32), ('ps', C.c_float)]


class RgStepResult(C.Structure):
    """struct rg_step_result: what rg_sim_step_user reads back."""
    _fields_ = [('row', RgEvent), ('state', C.c_int32), ('has_row', C.c_int32), ('time', C.c_double), ('ps', C.c_double),
                ('p_click', C.c_double)]


# every symbol include/recogym_hip.h declares, with its ctypes signature
_SIM = C.c_void_p
SYMBOLS = {
    'rg_last_error': (C.c_char_p, []),
    'rg_abi_version': (C.c_int, []),
    'rg_device_count': (C.c_int, []),
    'rg_sim_workspace_bytes': (C.c_size_t, [C.POINTER(RgConfig), C.c_uint64]),
    'rg_sim_create': (C.c_int, [C.POINTER(_SIM), C.POINTER(RgConfig), C.c_uint64, C.c_void_p,
                                C.c_size_t]),
    'rg_sim_destroy': (C.c_int, [_SIM]),
    'rg_sim_set_option': (C.c_int, [_SIM, C.c_char_p, C.c_int64]),
    'rg_sim_get_option': (C.c_int, [_SIM, C.c_char_p, C.POINTER(C.c_int64)]),
    'rg_sim_set_tables': (C.c_int, [_SIM, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p]),
    'rg_env0_click_thresholds': (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    'rg_sim_set_env0_tables': (C.c_int, [_SIM, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'rg_sim_set_policy_table': (C.c_int, [_SIM, C.c_void_p, C.c_void_p]),
    'rg_sim_set_logreg': (C.c_int, [_SIM, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]),
    'rg_sim_set_logreg_fp32': (C.c_int, [_SIM, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float]),
    'rg_sim_set_logreg_fp16': (C.c_int, [_SIM, C.c_void_p]),
    'rg_sim_set_logreg_int8': (C.c_int, [_SIM, C.c_void_p, C.c_void_p]),
    'rg_sim_set_log': (C.c_int, [_SIM, C.c_void_p, C.c_uint64]),
    'rg_sim_reset_users': (C.c_int, [_SIM, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p]),
    'rg_sim_reseed': (C.c_int, [_SIM, C.c_uint64, C.c_uint64]),
    'rg_sim_step': (C.c_int, [_SIM, C.c_void_p, C.c_void_p]),
    'rg_sim_step_user': (C.c_int, [_SIM, C.c_int32, C.POINTER(RgStepResult), C.c_void_p]),
    'rg_sim_run': (C.c_int, [_SIM, C.c_uint32, C.c_void_p]),
    'rg_sim_read_counters': (C.c_int, [_SIM, C.POINTER(C.c_int64), C.c_void_p]),
    'rg_sim_set_profiling': (C.c_int, [_SIM, C.c_int]),
    'rg_sim_get_profile': (C.c_int, [_SIM, C.POINTER(C.c_double)]),
    'rg_sim_export_state': (C.c_int, [_SIM, C.c_void_p, C.c_void_p]),
    'rg_sim_export_omega': (C.c_int, [_SIM, C.c_void_p, C.c_void_p]),
    'rg_sim_sort_log': (C.c_int, [_SIM, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                  C.c_void_p]),
    'rg_sim_set_log_aux': (C.c_int, [_SIM, C.c_void_p, C.c_void_p]),
    'rg_sim_sort_log_aux': (C.c_int, [_SIM, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                      C.c_void_p]),
    'rg_sim_set_log_time': (C.c_int, [_SIM, C.c_void_p]),
    'rg_sim_sort_log_time': (C.c_int, [_SIM, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    'rg_sim_export_time': (C.c_int, [_SIM, C.c_void_p, C.c_void_p]),
    'rg_sim_debug_set_omega': (C.c_int, [_SIM, C.c_void_p, C.c_void_p]),
    'rg_sim_debug_set_uniforms': (C.c_int, [_SIM, C.c_void_p]),
    'rg_sim_debug_set_row_base': (C.c_int, [_SIM, C.c_uint64]),
    'rg_sim_debug_uncertified': (C.c_int, [_SIM, C.c_void_p, C.c_void_p]),
    'rg_sim_debug_walk_fate': (C.c_int, [_SIM, C.c_void_p, C.c_void_p]),
    'rg_sim_debug_click_decisions': (C.c_int, [_SIM, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'rg_sim_debug_set_history': (C.c_int, [_SIM, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    'rg_sim_debug_ouc_acts': (C.c_int, [_SIM, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
}

LIB_NAME = 'librecogym_hip.so'
_lib = None


def lib_path():
    # RECOGYM_HIP_LIB: another build of the same source (A/B measurements of compile-time switches)
    return os.environ.get('RECOGYM_HIP_LIB') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc', LIB_NAME)


class RecoGymHipError(RuntimeError):
    pass


def load():
    """Load librecogym_hip.so (built by __graft_entry__.build()); raise loudly if absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise RecoGymHipError(
            f'{path} is missing: build it with `python -c "import __graft_entry__ as g; '
            f'g.build()"` (hipcc --offload-arch=gfx950). There is no CPU fallback.')
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)       # AttributeError here = header/library drift
        fn.restype = res
        fn.argtypes = args
    if lib.rg_abi_version() != RG_ABI_VERSION:
        raise RecoGymHipError(f'ABI mismatch: library {lib.rg_abi_version()} != {RG_ABI_VERSION}')
    _lib = lib
    return lib


def check(rc, what=''):
    if rc < 0:
        lib = load()
        msg = lib.rg_last_error()
        raise RecoGymHipError(
            f'{what}: {RG_ERRORS.get(rc, rc)}: {msg.decode() if msg else ""}')
    return rc
