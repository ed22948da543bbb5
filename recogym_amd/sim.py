"""Simulator — thin Python owner of one `rg_sim` handle (include/recogym_hip.h).

PyTorch-ROCm is plumbing here: it owns the device buffers (tables, workspace, log) and the
stream; every bit of simulation work happens in librecogym_hip.so's HIP kernels.  There is no
CPU fallback — constructing a Simulator without a HIP device raises.
"""
import ctypes as C
import math
import os

import numpy as np
import torch

from . import _abi
from .envs.static_params import draw_tables, make_rg_config

# decoded log rows (host side), same field names as the oracle's rows
ROW_DTYPE = np.dtype([('u', np.uint32), ('t', np.uint32), ('z', np.int32), ('v', np.int32),
                      ('a', np.int32), ('c', np.int32), ('phantom', np.int32),
                      ('ps', np.float64)])
ROW_DTYPE_PCLICK = np.dtype(ROW_DTYPE.descr + [('p_click', np.float64)])


def require_device(device=None):
    lib = _abi.load()
    if not torch.cuda.is_available() or lib.rg_device_count() <= 0:
        raise _abi.RecoGymHipError(
            'recogym_amd needs a HIP device (MI355X / gfx950); there is no CPU fallback')
    return torch.device(device if device is not None else f'cuda:{torch.cuda.current_device()}')


def default_log_capacity(config, n_users):
    """Rows a run emits: each user lives ~Geometric(prob_leave) events plus its phantom row."""
    p_stop = max(float(config.prob_leave_organic), 1e-6)
    mean = 1.0 / p_stop + 1.0
    rows = n_users * (mean + 1.0) + 8.0 * mean * math.sqrt(n_users) + 4096
    # the user-major walk (sigma_omega = 0) reserves raw-log rows per wave in chunks and leaves a few percent of
    # them unused: up to 63 per chunk of >= 256 rows and the rest of every wave's last chunk
    return int(rows * 1.04 + min(rows * 0.3, 4.0e6) + 65536)


def decode_rows(raw, uniform_ps=None, ps64=None, p_click=None):
    """(n,4) int32 array of rg_event records -> structured host rows.  `ps64` / `p_click`: the float64
    side arrays in the same order (Simulator.sorted_aux_host), else ps is the row's float32 value."""
    raw = np.ascontiguousarray(raw).view(np.uint32).reshape(-1, 4)
    n = raw.shape[0]
    out = np.zeros(n, dtype=ROW_DTYPE if p_click is None else ROW_DTYPE_PCLICK)
    code = raw[:, 2]
    is_b = (code & _abi.RG_EV_BANDIT) != 0
    idx = (code & _abi.RG_EV_INDEX_MASK).astype(np.int32)
    out['u'] = raw[:, 0]
    out['t'] = raw[:, 1]
    out['z'] = is_b
    out['v'] = np.where(is_b, -1, idx)
    out['a'] = np.where(is_b, idx, -1)
    out['c'] = np.where(is_b, (code & _abi.RG_EV_CLICK) != 0, -1)
    out['phantom'] = (code & _abi.RG_EV_PHANTOM) != 0
    ps = raw[:, 3].copy().view(np.float32).astype(np.float64) if ps64 is None else ps64
    if p_click is not None:
        out['p_click'] = np.where(is_b & (out['phantom'] == 0), p_click, np.nan)
    if uniform_ps is not None:
        ps = np.where(is_b, uniform_ps, np.nan)     # exact 1/P of the uniform policies
    out['ps'] = np.where(is_b, ps, np.nan)
    return out


class Simulator:
    """N concurrent users of one reco-gym-v1 environment on one GPU.

    config       env Configuration (env_1_args keys)
    n_users      capacity (users per reset range)
    policy       _abi.RG_POLICY_*; policy_seed / ouc = the agent's parameters
    epoch        added to config.random_seed (reset_random_seed(epoch), abstract.py:59-62)
    log_capacity rows of device log to allocate; 0 = counters only; None = estimate
    ps_float64   keep the propensity of every bandit row in float64 beside the 16-byte row (the dtype of
                 the reference's `ps` column); default: on for the policies whose ps is not a constant
    p_click      also keep the click probability of every bandit row (parity checks)
    """

    def __init__(self, config, n_users, policy=_abi.RG_POLICY_UNIFORM_ENV, policy_seed=None,
                 ouc=None, epoch=0, log_capacity=None, device=None, tables=None, policy_table=None,
                 policy_ps=None, logreg=None, ps_float64=None, p_click=False, env0=None):
        self.lib = _abi.load()
        self.device = require_device(device)
        self.config = config
        self.n_users = int(n_users)
        # env0: the dict draw_env0_tables(config) returns — reco-gym-v0 (env_kind = 1), every draw a table look-up
        self.env0 = env0
        self.rg_config = make_rg_config(config, config.random_seed + epoch, policy, policy_seed,
                                        ouc, env_kind=1 if env0 is not None else 0,
                                        lr_select_randomly=bool(logreg and logreg.get('select_randomly')))
        self.policy = policy
        self.time_mode = int(self.rg_config.time_mode)
        self.ps_float64 = ((policy in (_abi.RG_POLICY_ORGANIC_USER_COUNT, _abi.RG_POLICY_LAST_VIEW_TABLE)
                            or bool(logreg and logreg.get('select_randomly')))
                           if ps_float64 is None else bool(ps_float64))
        self.keep_p_click = bool(p_click)
        host_tables = () if env0 is not None else (tables if tables is not None else draw_tables(config))
        self.host_tables = host_tables
        with torch.cuda.device(self.device):
            self.tables = [torch.from_numpy(np.ascontiguousarray(t)).to(self.device)
                           for t in host_tables]
            need = self.lib.rg_sim_workspace_bytes(C.byref(self.rg_config), self.n_users)
            if need == 0:
                raise _abi.RecoGymHipError('rg_sim_workspace_bytes: ' +
                                           self.lib.rg_last_error().decode())
            self.workspace = self._poison(torch.empty(need, dtype=torch.uint8, device=self.device))
            self._h = C.c_void_p()
            _abi.check(self.lib.rg_sim_create(C.byref(self._h), C.byref(self.rg_config),
                                              self.n_users, self.workspace.data_ptr(), need),
                       'rg_sim_create')
            if env0 is not None:
                p = np.ascontiguousarray(env0['click_probs'], dtype=np.float64)
                qn, px1 = np.empty_like(p), np.empty_like(p)
                _abi.check(self.lib.rg_env0_click_thresholds(p.ctypes.data, p.size, qn.ctypes.data, px1.ctypes.data),
                           'rg_env0_click_thresholds')
                self.tables = [torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64)).to(self.device)
                               for x in (env0['cdf_init'], env0['cdf_cluster'], p, qn, px1)]
                _abi.check(self.lib.rg_sim_set_env0_tables(self._h, self.tables[0].data_ptr(), self.tables[1].data_ptr(),
                                                           int(env0['cluster_size']), self.tables[2].data_ptr(),
                                                           self.tables[3].data_ptr(), self.tables[4].data_ptr(), self._stream()),
                           'rg_sim_set_env0_tables')
            else:
                _abi.check(self.lib.rg_sim_set_tables(self._h, *[t.data_ptr() for t in self.tables],
                                                      self._stream()), 'rg_sim_set_tables')
            if policy == _abi.RG_POLICY_LAST_VIEW_TABLE:
                self.policy_table = torch.as_tensor(np.ascontiguousarray(policy_table, dtype=np.int32)).to(self.device)
                self.policy_ps = None if policy_ps is None else \
                    torch.as_tensor(np.ascontiguousarray(policy_ps, dtype=np.float32)).to(self.device)
                _abi.check(self.lib.rg_sim_set_policy_table(
                    self._h, self.policy_table.data_ptr(),
                    None if self.policy_ps is None else self.policy_ps.data_ptr()), 'rg_sim_set_policy_table')
            if policy == _abi.RG_POLICY_LOGREG_FROZEN:
                # dict(coef_t (P, C) float64 = sklearn coef_.T, intercept (C,), classes (C,))
                n_fit = int(np.asarray(logreg['classes']).size)
                if n_fit % 8 and logreg.get('fp16', True) and logreg.get('fp32', True) and not logreg.get('select_randomly'):
                    # the fp16 screening pass reads 8 classes per 16-byte load: pad with copies of the LAST class (same
                    # column, intercept and action: an exact tie with it, whichever of them wins the action is the same)
                    pad = 8 - n_fit % 8
                    logreg = dict(logreg,
                                  coef_t=np.concatenate([logreg['coef_t'], np.repeat(np.asarray(logreg['coef_t'])[:, -1:], pad, axis=1)], axis=1),
                                  intercept=np.concatenate([logreg['intercept'], np.repeat(np.asarray(logreg['intercept'])[-1:], pad)]),
                                  classes=np.concatenate([logreg['classes'], np.repeat(np.asarray(logreg['classes'])[-1:], pad)]))
                self.logreg = (
                    torch.as_tensor(np.ascontiguousarray(logreg['coef_t'], dtype=np.float64)).to(self.device),
                    torch.as_tensor(np.ascontiguousarray(logreg['intercept'], dtype=np.float64)).to(self.device),
                    torch.as_tensor(np.ascontiguousarray(logreg['classes'], dtype=np.int32)).to(self.device))
                assert self.logreg[0].shape == (config.num_products, self.logreg[2].numel())
                _abi.check(self.lib.rg_sim_set_logreg(self._h, self.logreg[0].data_ptr(), self.logreg[1].data_ptr(),
                                                      self.logreg[2].data_ptr(), self.logreg[2].numel()),
                           'rg_sim_set_logreg')
                if logreg.get('fp32', True) and not logreg.get('select_randomly'):
                    # fp32 copies for the certified fast scores (the float64 arrays stay the arbiter); bounds rounded up
                    w32 = self.logreg[0].to(torch.float32)
                    b32 = self.logreg[1].to(torch.float32)
                    wmax = (self.logreg[0].abs().amax(dim=1) * (1.0 + 1e-6)).to(torch.float32) * (1.0 + 1e-6)
                    bmax = float(self.logreg[1].abs().max().item()) * (1.0 + 1e-6)
                    self.logreg32 = (w32.contiguous(), b32.contiguous(), wmax.contiguous())
                    _abi.check(self.lib.rg_sim_set_logreg_fp32(self._h, self.logreg32[0].data_ptr(), self.logreg32[1].data_ptr(),
                                                               self.logreg32[2].data_ptr(), C.c_float(bmax)), 'rg_sim_set_logreg_fp32')
                    # screening pass from a half copy of coef^T (half the row bytes; float64 decides among the candidates):
                    # RECOGYM_LOGREG=fp32 keeps the fp32 scores only (A/B)
                    n_cls = int(self.logreg[2].numel())
                    if (logreg.get('fp16', True) and n_cls % 8 == 0 and os.environ.get('RECOGYM_LOGREG', 'fp16') != 'fp32'
                            and float(self.logreg[0].abs().max().item()) < 6.0e4):
                        self.logreg16 = self.logreg[0].to(torch.float16).contiguous()
                        _abi.check(self.lib.rg_sim_set_logreg_fp16(self._h, self.logreg16.data_ptr()), 'rg_sim_set_logreg_fp16')
                        # round 6, opt-in (RECOGYM_LOGREG=int8 or logreg['int8']): the screening pass from an 8-BIT copy (a quarter of the
                        # fp32 row bytes): q = rint(w / scale) + 128, scale = wmax / 127 per product row — |w - scale q| <= scale / 2.
                        # Its band is 8x the fp16 copy's, so the classes it keeps are scored once more from the fp16 rows (second level,
                        # inside the screen kernel) before float64 decides
                        if logreg.get('int8', False) or os.environ.get('RECOGYM_LOGREG', 'fp16') == 'int8':
                            scale = (wmax.to(torch.float64) / 127.0 * (1.0 + 1e-6)).clamp_min(1e-30)
                            q = torch.round(self.logreg[0] / scale[:, None]).clamp_(-127, 127)
                            assert bool(((self.logreg[0] - q * scale[:, None]).abs() <= 0.5 * scale[:, None] * (1.0 + 1e-9)).all())
                            # (16 bytes of padding behind the last row: the pass reads 20 bytes per lane and row)
                            q8 = torch.zeros(q.numel() + 16, dtype=torch.uint8, device=q.device)
                            q8[:q.numel()] = (q + 128.0).to(torch.uint8).reshape(-1)
                            self.logreg8 = (q8, scale.to(torch.float32).contiguous())
                            # (the float32 scale the kernel multiplies with must not be SMALLER than the one the weights were divided
                            # by, or the half-step bound would be off by the rounding: compare against the float64 value)
                            assert bool((self.logreg8[1].to(torch.float64) * (1.0 + 2e-7) >= scale).all())
                            _abi.check(self.lib.rg_sim_set_logreg_int8(self._h, self.logreg8[0].data_ptr(), self.logreg8[1].data_ptr()),
                                       'rg_sim_set_logreg_int8')
            if log_capacity is None:
                log_capacity = default_log_capacity(config, self.n_users)
            self.log = None
            self.set_log_capacity(log_capacity)

    # -- plumbing ---------------------------------------------------------------------------
    @staticmethod
    def _poison(t):
        """RECOGYM_POISON=1 (a debugging aid): fill every buffer handed to the library with 0xA5 bytes, so that a
        kernel reading memory nothing initialised fails every time instead of when the allocator recycles a block."""
        if os.environ.get('RECOGYM_POISON'):
            t.view(torch.uint8).fill_(0xA5)
        return t

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def set_log_capacity(self, rows):
        self.log_capacity = int(rows)
        with torch.cuda.device(self.device):
            self.log = (self._poison(torch.empty((self.log_capacity, 4), dtype=torch.int32, device=self.device))
                        if rows else None)
        _abi.check(self.lib.rg_sim_set_log(self._h, self.log.data_ptr() if rows else None,
                                           self.log_capacity), 'rg_sim_set_log')
        self.aux_time = None
        if rows and self.time_mode:
            with torch.cuda.device(self.device):
                self.aux_time = self._poison(torch.empty(self.log_capacity, dtype=torch.float64, device=self.device))
            _abi.check(self.lib.rg_sim_set_log_time(self._h, self.aux_time.data_ptr()), 'rg_sim_set_log_time')
        self.aux_ps = self.aux_p_click = None
        if rows and (self.ps_float64 or self.keep_p_click):
            with torch.cuda.device(self.device):
                if self.ps_float64:
                    self.aux_ps = self._poison(torch.empty(self.log_capacity, dtype=torch.float64, device=self.device))
                if self.keep_p_click:
                    self.aux_p_click = self._poison(torch.empty(self.log_capacity, dtype=torch.float64, device=self.device))
            _abi.check(self.lib.rg_sim_set_log_aux(
                self._h, None if self.aux_ps is None else self.aux_ps.data_ptr(),
                None if self.aux_p_click is None else self.aux_p_click.data_ptr()), 'rg_sim_set_log_aux')

    def close(self):
        if getattr(self, '_h', None) is not None and self._h:
            self.lib.rg_sim_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- the step loop ----------------------------------------------------------------------
    def reseed(self, seed, policy_seed=None):
        ps = self.rg_config.policy_seed if policy_seed is None else policy_seed
        self.rg_config.seed = seed
        self.rg_config.policy_seed = ps
        _abi.check(self.lib.rg_sim_reseed(self._h, seed, ps), 'rg_sim_reseed')

    def set_option(self, name, value):
        """A run-path tuning knob by name (rg_sim_set_option; include/recogym_hip.h lists them)."""
        _abi.check(self.lib.rg_sim_set_option(self._h, name.encode(), int(value)), f'rg_sim_set_option({name})')

    def get_option(self, name):
        v = C.c_int64(0)
        _abi.check(self.lib.rg_sim_get_option(self._h, name.encode(), C.byref(v)), f'rg_sim_get_option({name})')
        return int(v.value)

    def reset_users(self, first_user_id=0, n=None, organic_only_below=0):
        n = self.n_users if n is None else int(n)
        self.first_user_id = int(first_user_id)
        self.n_active = n
        with torch.cuda.device(self.device):
            _abi.check(self.lib.rg_sim_reset_users(self._h, self.first_user_id, n,
                                                   int(organic_only_below), self._stream()),
                       'rg_sim_reset_users')

    def step(self, actions=None):
        """One Markov transition for every live user; `actions` (int32 tensor, one per user of
        the reset range) only with the external policy."""
        ptr = None
        if actions is not None:
            assert actions.dtype == torch.int32 and actions.is_cuda
            ptr = actions.data_ptr()
        with torch.cuda.device(self.device):
            _abi.check(self.lib.rg_sim_step(self._h, ptr, self._stream()), 'rg_sim_step')

    def step_user(self, action=None):
        """One Markov transition of a ONE-user external-policy simulator with a single read-back (rg_sim_step_user) ->
        (decoded row or None, state, clock)."""
        res = _abi.RgStepResult()
        with torch.cuda.device(self.device):
            _abi.check(self.lib.rg_sim_step_user(self._h, -1 if action is None else int(action), C.byref(res), self._stream()),
                       'rg_sim_step_user')
        row = None
        if res.has_row:
            raw = np.array([[res.row.u, res.row.t, res.row.code, 0]], dtype=np.uint32)
            raw[0, 3] = np.float32(res.row.ps).view(np.uint32)
            row = decode_rows(raw.view(np.int32), ps64=np.array([res.ps]) if self.aux_ps is not None else None,
                              p_click=np.array([res.p_click]) if self.aux_p_click is not None else None)[0]
        return row, int(res.state), float(res.time)

    def run(self, max_steps=1 << 16):
        """rg_sim_run; raises when the run is incomplete (uncertified draws beyond the float64 resolve scratch)."""
        with torch.cuda.device(self.device):
            _abi.check(self.lib.rg_sim_run(self._h, max_steps, self._stream()), 'rg_sim_run')

    def counters(self):
        out = (C.c_int64 * _abi.RG_CNT_N)()
        with torch.cuda.device(self.device):
            _abi.check(self.lib.rg_sim_read_counters(self._h, out, self._stream()),
                       'rg_sim_read_counters')
        names = ['organic', 'bandit', 'clicks', 'phantom', 'live', 'step', 'log_rows',
                 'log_dropped', 'exact_draws', 'hist_overflow', 'exact_sweeps', 'exact_overflow', 'lr_acts', 'lr_rows',
                 'lr_exact', 'memo_hits']
        res = {k: int(out[i]) for i, k in enumerate(names)}
        res['anchored'] = int(out[_abi.RG_CNT_ANCHORED])
        res['bad_actions'] = int(out[_abi.RG_CNT_BAD_ACTION])        # external actions outside [0, P) that reached the device
        return res

    def set_profiling(self, on=True):
        _abi.check(self.lib.rg_sim_set_profiling(self._h, int(on)), 'rg_sim_set_profiling')

    def profile(self):
        """-> dict(draw_mfma_ms, draw_search_ms, draw_exact_ms, advance_ms, steps, tail_ms), HIP events."""
        out = (C.c_double * 10)()
        _abi.check(self.lib.rg_sim_get_profile(self._h, out), 'rg_sim_get_profile')
        return dict(draw_mfma_ms=out[0], draw_search_ms=out[1], draw_exact_ms=out[2],
                    advance_ms=out[3], steps=int(out[4]), tail_ms=out[5], walk1_ms=out[6], walk2_ms=out[7],
                    logreg_ms=out[8])

    def states(self):
        with torch.cuda.device(self.device):
            st = torch.empty(self.n_users, dtype=torch.int8, device=self.device)
            _abi.check(self.lib.rg_sim_export_state(self._h, st.data_ptr(), self._stream()),
                       'rg_sim_export_state')
        return st[:self.n_active]

    def walk_fate(self):
        """uint8 per user of the reset range after a walked run: bit 0 = went through the float64 batch and round 2, bit 1 =
        finished by the last round (rg_sim_debug_walk_fate; the sampled-oracle check picks its users with it)."""
        with torch.cuda.device(self.device):
            fl = torch.zeros(self.n_users, dtype=torch.uint8, device=self.device)
            _abi.check(self.lib.rg_sim_debug_walk_fate(self._h, fl.data_ptr(), self._stream()), 'rg_sim_debug_walk_fate')
        return fl[:self.n_active]

    def omega(self):
        with torch.cuda.device(self.device):
            om = torch.empty((self.n_active, self.config.K), dtype=torch.float64,
                             device=self.device)
            _abi.check(self.lib.rg_sim_export_omega(self._h, om.data_ptr(), self._stream()),
                       'rg_sim_export_omega')
        return om

    # -- logs -------------------------------------------------------------------------------
    def sorted_log(self):
        """Device log in the reference's row order -> (rows,4) int32 tensor + int64 offsets."""
        if self.log is None:
            raise _abi.RecoGymHipError('this Simulator keeps counters only (log_capacity=0)')
        n = self.n_active
        with torch.cuda.device(self.device):
            offsets = torch.empty(n + 1, dtype=torch.int64, device=self.device)
            scratch = torch.empty(n + (n + 255) // 256 + 1, dtype=torch.int64, device=self.device)
            cnt = self.counters()
            total = cnt['organic'] + cnt['bandit'] + cnt['phantom']
            if cnt['log_dropped']:
                raise _abi.RecoGymHipError(
                    f'log overflow: capacity {self.log_capacity}, '
                    f'{cnt["log_rows"] + cnt["log_dropped"]} rows emitted')
            out = torch.empty((total, 4), dtype=torch.int32, device=self.device)
            _abi.check(self.lib.rg_sim_sort_log(self._h, offsets.data_ptr(), scratch.data_ptr(),
                                                out.data_ptr(), total, self._stream()),
                       'rg_sim_sort_log')
        return out, offsets

    def sorted_time(self, offsets, total):
        """The time column in the order of sorted_log() (float64): the clock of a NormalTimeGenerator, else the event index."""
        with torch.cuda.device(self.device):
            out = torch.empty(total, dtype=torch.float64, device=self.device)
            _abi.check(self.lib.rg_sim_sort_log_time(self._h, offsets.data_ptr(), out.data_ptr(), total, self._stream()),
                       'rg_sim_sort_log_time')
        return out

    def user_times(self):
        """Current clock of every user of the reset range (float64 device tensor)."""
        with torch.cuda.device(self.device):
            out = torch.empty(self.n_active, dtype=torch.float64, device=self.device)
            _abi.check(self.lib.rg_sim_export_time(self._h, out.data_ptr(), self._stream()), 'rg_sim_export_time')
        return out

    def sorted_aux(self, offsets, total):
        """The float64 side arrays (ps, p_click) in the order of sorted_log(); None where not kept."""
        with torch.cuda.device(self.device):
            ps = None if self.aux_ps is None else torch.empty(total, dtype=torch.float64, device=self.device)
            pc = None if self.aux_p_click is None else torch.empty(total, dtype=torch.float64, device=self.device)
            if ps is not None or pc is not None:
                _abi.check(self.lib.rg_sim_sort_log_aux(
                    self._h, offsets.data_ptr(), None if ps is None else ps.data_ptr(),
                    None if pc is None else pc.data_ptr(), total, self._stream()), 'rg_sim_sort_log_aux')
        return ps, pc

    def sorted_log_host(self):
        """-> ((n, 4) int32 host array of the log in the reference's row order, exact `ps` of the
        uniform policies or None)."""
        out, _ = self.sorted_log()
        uniform = None
        if self.policy in (_abi.RG_POLICY_UNIFORM_ENV, _abi.RG_POLICY_RANDOM_AGENT):
            uniform = 1.0 / float(self.config.num_products)
        return out.cpu().numpy(), uniform

    def log_columns(self, on_device=False, chunk_rows=1 << 25, copy=True):
        """The log in the reference's row order, decoded ON THE DEVICE into the columns of the reference's DataFrame
        (SURVEY.md §8f-2): dict(t f32, u i32, is_bandit bool, v i32, a i32, c f32 (NaN on organic rows), ps f64 (NaN)).

        To the host (the default) the columns travel in chunks of `chunk_rows` rows: chunk i is decoded into one of two staging
        sets on the simulator's stream while chunk i - 1 crosses PCIe on a copy stream, straight into PINNED host buffers the
        simulator keeps between calls (allocated on the first call, grown when a log is longer).  `copy=True` (the default)
        returns arrays that OWN their memory; `copy=False` returns zero-copy NumPy views of those pinned buffers — valid only
        until the next log_columns() of this simulator, which reuses them, and page-locked for as long as anything references them
        (for callers that control that lifetime: bench.py's materialise figure, a DataFrame built and dropped at once)."""
        out, offsets = self.sorted_log()
        ps64, _ = self.sorted_aux(offsets, out.shape[0])
        n = int(out.shape[0])
        uniform = self.policy in (_abi.RG_POLICY_UNIFORM_ENV, _abi.RG_POLICY_RANDOM_AGENT)
        t_sorted = self.sorted_time(offsets, n) if self.time_mode else None

        def decode(lo, hi, dst=None):
            """columns of rows [lo, hi) as device tensors (into the staging set `dst` when given)"""
            rows = out[lo:hi]
            code = rows[:, 2]
            is_b = (code & _abi.RG_EV_BANDIT) != 0
            idx = code & _abi.RG_EV_INDEX_MASK
            zero = torch.zeros((), dtype=torch.int32, device=out.device)
            nan32 = torch.full((), float('nan'), dtype=torch.float32, device=out.device)
            nan64 = torch.full((), float('nan'), dtype=torch.float64, device=out.device)
            if uniform:
                ps_b = torch.full((), 1.0 / float(self.config.num_products), dtype=torch.float64, device=out.device)   # exact 1/P
            elif ps64 is not None:
                ps_b = ps64[lo:hi]                                  # float64 as the policy computed it
            else:
                ps_b = rows[:, 3].contiguous().view(torch.float32).to(torch.float64)
            cols = dict(
                t=(t_sorted[lo:hi] if t_sorted is not None else rows[:, 1]).to(torch.float32),
                u=rows[:, 0].contiguous(),
                is_bandit=is_b,
                v=torch.where(is_b, zero, idx),
                a=torch.where(is_b, idx, zero),
                c=torch.where(is_b, ((code & _abi.RG_EV_CLICK) != 0).to(torch.float32), nan32),
                ps=torch.where(is_b, ps_b, nan64),
            )
            if dst is not None:
                for k, v in cols.items():
                    dst[k][:hi - lo].copy_(v)
                return dst
            return cols

        with torch.cuda.device(self.device):
            if on_device:
                return decode(0, n)
            host = self._host_columns(n)
            if n == 0:
                return {k: v[:0].numpy() for k, v in host.items()}
            chunk = int(max(1, min(chunk_rows, n)))
            if getattr(self, '_copy_stream', None) is None:
                self._copy_stream = torch.cuda.Stream(device=self.device)
            cs = self._copy_stream
            stage = getattr(self, '_stage_cols', None)
            if stage is None or stage[0]['u'].shape[0] < chunk:
                stage = self._stage_cols = [{k: torch.empty(chunk, dtype=v.dtype, device=self.device) for k, v in host.items()}
                                            for _ in range(2)]
            done = [None, None]
            main = torch.cuda.current_stream(self.device)
            for i, lo in enumerate(range(0, n, chunk)):
                hi = min(lo + chunk, n)
                if done[i & 1] is not None:
                    main.wait_event(done[i & 1])        # the copy that last read this staging set has finished
                decode(lo, hi, stage[i & 1])
                ready = torch.cuda.Event()
                ready.record(main)
                cs.wait_event(ready)
                with torch.cuda.stream(cs):
                    for k, v in host.items():
                        v[lo:hi].copy_(stage[i & 1][k][:hi - lo], non_blocking=True)
                    done[i & 1] = torch.cuda.Event()
                    done[i & 1].record(cs)
            cs.synchronize()
            return {k: (v[:n].numpy().copy() if copy else v[:n].numpy()) for k, v in host.items()}

    _HOST_COLS = (('t', torch.float32), ('u', torch.int32), ('is_bandit', torch.bool), ('v', torch.int32), ('a', torch.int32),
                  ('c', torch.float32), ('ps', torch.float64))

    def _host_columns(self, n):
        """Host buffers of log_columns(): pinned (page-locked: PCIe DMA at full rate, asynchronous copies), kept between calls.
        Where the host cannot spare the memory page-locked (less than twice the need is available) they are ordinary arrays."""
        have = getattr(self, '_host_cols', None)
        if have is not None and have['u'].shape[0] >= n:
            return have
        cap = int(n * 1.02) + 4096
        need = cap * sum(torch.empty((), dtype=d).element_size() for _, d in self._HOST_COLS)
        pin = True
        try:
            avail = next(int(l.split()[1]) * 1024 for l in open('/proc/meminfo') if l.startswith('MemAvailable'))
            pin = need * 2 < avail
        except Exception:       # noqa: BLE001 — no /proc/meminfo: pin
            pass
        self._host_cols = None
        self._host_cols = {k: torch.empty(cap, dtype=d, pin_memory=pin) for k, d in self._HOST_COLS}
        self._host_cols_pinned = pin
        return self._host_cols

    def log_columns_device(self):
        """log_columns() left on the device (torch tensors), e.g. for
        agents.feature_feed.train_data_from_log_torch."""
        return self.log_columns(on_device=True)

    def raw_log(self):
        """The raw (unsorted) device log without its unused entries: (rows, 4) int32 tensor.  The user-major
        walk reserves rows per wave in chunks and marks what it leaves unused (code 0xFFFFFFFF)."""
        n = min(self.counters()['log_rows'], self.log_capacity)
        raw = self.log[:n]
        if not bool((raw[:, 2] == -1).any()):
            return raw                                  # lock-step runs leave no unused entries: a view, no copy
        # piecewise: torch's masked select mis-indexes results beyond 2^31 bytes on this stack
        step = 1 << 24
        return torch.cat([raw[i:i + step][raw[i:i + step, 2] != -1] for i in range(0, n, step)])

    @staticmethod
    def _mix64(x):
        """splitmix64 finaliser on int64 tensors (arithmetic wraps modulo 2^64; shifts made logical by masking)."""
        def shr(v, k):
            return (v >> k) & ((1 << (64 - k)) - 1)
        x = x + (-7046029254386353131)                     # 0x9E3779B97F4A7C15
        x = (x ^ shr(x, 30)) * (-4658895280553007687)      # 0xBF58476D1CE4E5B9
        x = (x ^ shr(x, 27)) * (-7723592293110705685)      # 0x94D049BB133111EB
        return x ^ shr(x, 31)

    def log_digest(self, phantom=True):
        """Order-independent checksum of the run's log: one sum (modulo 2^64) per column of the 16-byte row (u, t, code,
        ps bits) and one over the float64 `ps` side array's bit patterns, every row weighted by a 64-bit hash of its key
        (u, t) — a key occurs once per run, so values swapped between rows do not cancel.  Unused entries of a walked run
        are skipped.  `phantom`: the trailing phantom row of every user (kept beside the raw log) is folded in too, which
        costs one rg_sim_sort_log (the digest is then taken over the sorted log).  Equal digests of two runs = a checksum
        match of the same multiset of rows, whatever order the execution form emitted them in; digests of disjoint user
        shards add up (mod 2^64) to the digest of the whole run."""
        if phantom:
            rows_all, offsets = self.sorted_log()
            ps64, _ = self.sorted_aux(offsets, rows_all.shape[0])
            n = int(rows_all.shape[0])
        else:
            n = min(self.counters()['log_rows'], self.log_capacity)
            rows_all, ps64 = self.log, self.aux_ps
        chk = [0, 0, 0, 0, 0]
        step = 1 << 25
        for lo in range(0, n, step):
            rows = rows_all[lo:min(lo + step, n)].to(torch.int64)
            live = (rows[:, 2] != -1)
            w = self._mix64((rows[:, 0] << 32) | (rows[:, 1] & 0xFFFFFFFF)) * live.to(torch.int64)
            for i in range(4):
                chk[i] = (chk[i] + int((rows[:, i] * w).sum().item())) % (1 << 64)
            if ps64 is not None:
                is_b = live & ((rows[:, 2] & _abi.RG_EV_BANDIT) != 0)
                bits = ps64[lo:min(lo + step, n)].view(torch.int64)
                chk[4] = (chk[4] + int((torch.where(is_b, bits, torch.zeros_like(bits)) * w).sum().item())) % (1 << 64)
        return chk

    def rows(self):
        """Decoded host rows in the reference's order."""
        out, offsets = self.sorted_log()
        ps64, pc = self.sorted_aux(offsets, out.shape[0])
        uniform = None
        if self.policy in (_abi.RG_POLICY_UNIFORM_ENV, _abi.RG_POLICY_RANDOM_AGENT):
            uniform = 1.0 / float(self.config.num_products)
        rows = decode_rows(out.cpu().numpy(), uniform, None if ps64 is None else ps64.cpu().numpy(),
                           None if pc is None else pc.cpu().numpy())
        if self.time_mode:         # 't' stays the event index; 'time' = the generator's clock
            import numpy.lib.recfunctions as rfn
            rows = rfn.append_fields(rows, 'time', self.sorted_time(offsets, out.shape[0]).cpu().numpy(), usemask=False)
        return rows
