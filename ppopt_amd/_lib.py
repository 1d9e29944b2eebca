"""ctypes binding of libmpcombi_hip.so (the C ABI declared in include/mpcombi.h).

This is the only doorway from the Python host code to the device kernels.  There is no CPU fallback: if the
shared library is missing, or no HIP device is usable, every call raises ``MpcError``.
"""
import ctypes
import weakref
import os
from typing import Optional

import numpy

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('MPC_LIB_PATH') or os.path.join(_HERE, 'csrc', 'libmpcombi_hip.so')   # MPC_LIB_PATH: A/B builds

MPC_OK, MPC_ERR_INVALID, MPC_ERR_HIP, MPC_ERR_CAPACITY, MPC_ERR_STATE = range(5)
MPC_LOCATE_OVERLAPPING, MPC_LOCATE_INCLUSIVE, MPC_LOCATE_WALK = 1, 2, 4   # flags of mpc_locator_query
MPC_SOLVE_MANY_BASE = 128   # flag of mpc_solve_many_start
MPC_LEVEL_STREAM, MPC_LEVEL_GRAPH, MPC_LEVEL_THEN_BASE, MPC_LEVEL_KEEP_LOWDIM, MPC_LEVEL_ONLY_BASE = 1, 4, 8, 16, 32   # flags of mpc_level_start / mpc_level_run_ex
MPC_SOLVE_FETCH = 64   # flag of mpc_solve_start
INFEASIBLE, FEASIBLE, OPTIMAL_NO_REGION, REGION, SINGULAR_KKT, LP_LIMIT = range(6)
LP_OPTIMAL, LP_INFEASIBLE, LP_UNBOUNDED, LP_ITERLIMIT = range(4)
MASK_WORDS = 2

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int32)
_lp = ctypes.POINTER(ctypes.c_int64)
_u8p = ctypes.POINTER(ctypes.c_uint8)
_u64p = ctypes.POINTER(ctypes.c_uint64)


class MpcError(RuntimeError):
    code = 0


class MpcCapacityError(MpcError):
    """MPC_ERR_CAPACITY from a level: the overlapped region stage found more late optimal candidates than it had reserved
    slots for (mpcombi.h, mpc_set_region_overlap).  The drivers repeat the solve with the overlap switched off."""
    code = 3


class MpcProblem(ctypes.Structure):
    _fields_ = [('n_x', ctypes.c_int32), ('n_t', ctypes.c_int32), ('n_c', ctypes.c_int32), ('n_eq', ctypes.c_int32),
                ('n_tc', ctypes.c_int32), ('A', _dp), ('b', _dp), ('F', _dp), ('c', _dp), ('H', _dp), ('Q', _dp),
                ('A_t', _dp), ('b_t', _dp)]


class LevelStats(ctypes.Structure):
    _fields_ = [('n', ctypes.c_int64), ('k', ctypes.c_int32), ('kkt_mode', ctypes.c_int32),
                ('n_status', ctypes.c_int64 * 6), ('n_regions', ctypes.c_int64), ('n_children', ctypes.c_int64),
                ('n_pruned_new', ctypes.c_int64), ('lp_pivots', ctypes.c_int64), ('ms_verdict', ctypes.c_float),
                ('ms_region', ctypes.c_float), ('ms_children', ctypes.c_float), ('ms_total', ctypes.c_float),
                ('n_xtheta_lp', ctypes.c_int64), ('n_xtheta_fallback', ctypes.c_int64),
                ('wave_cycles', ctypes.c_int64 * 4), ('n_region_retry', ctypes.c_int64),
                ('n_x_cached', ctypes.c_int64), ('ms_theta', ctypes.c_float), ('ms_x', ctypes.c_float),
                ('ms_region2', ctypes.c_float), ('region_side_stream', ctypes.c_float), ('n_x_items', ctypes.c_int64),
                ('n_opt', ctypes.c_int64), ('dict_read_bytes', ctypes.c_int64), ('dict_write_bytes', ctypes.c_int64),
                ('n_theta_items', ctypes.c_int64), ('n_region_rows', ctypes.c_int64),
                ('ms_kkt', ctypes.c_float), ('ms_xq', ctypes.c_float), ('n_xq_items', ctypes.c_int64), ('xq_pivots', ctypes.c_int64),
                ('xq_record_ints', ctypes.c_int64), ('xq_record_rows', ctypes.c_int64), ('xq_record_cols', ctypes.c_int64),
                ('n_xq_thread', ctypes.c_int64), ('ms_xq_thread', ctypes.c_float), ('xq_thread_beside_theta', ctypes.c_float),
                ('n_x1', ctypes.c_int64), ('ms_x1', ctypes.c_float), ('ms_x_plan', ctypes.c_float)]


class SolveLevelInfo(ctypes.Structure):
    """mpc_solve_level_info (include/mpcombi.h)."""
    _fields_ = [('level', ctypes.c_int32), ('k', ctypes.c_int32), ('mode', ctypes.c_int32), ('chunk', ctypes.c_int32),
                ('n_chunks', ctypes.c_int32), ('pad_', ctypes.c_int32), ('n', ctypes.c_int64), ('n_slots', ctypes.c_int64),
                ('n_rows', ctypes.c_int64), ('head_d', ctypes.c_void_p), ('head_i', ctypes.c_void_p), ('erows', ctypes.c_void_p)]


class ManyLevelInfo(ctypes.Structure):
    """mpc_many_level_info (include/mpcombi.h)."""
    _fields_ = [('level', ctypes.c_int32), ('n_members', ctypes.c_int32), ('n_shared', ctypes.c_int32), ('done', ctypes.c_int32),
                ('base', ctypes.c_int32), ('pad_', ctypes.c_int32),
                ('member', ctypes.POINTER(ctypes.c_int32)), ('stats', ctypes.POINTER(LevelStats)),
                ('n_slots', ctypes.POINTER(ctypes.c_int64)), ('n_rows', ctypes.POINTER(ctypes.c_int64)),
                ('off_d', ctypes.POINTER(ctypes.c_int64)), ('off_i', ctypes.POINTER(ctypes.c_int64)), ('off_e', ctypes.POINTER(ctypes.c_int64)),
                ('head_d', ctypes.c_void_p), ('head_i', ctypes.c_void_p), ('erows', ctypes.c_void_p),
                ('len_d', ctypes.c_int64), ('len_i', ctypes.c_int64), ('len_e', ctypes.c_int64), ('ms_wall', ctypes.c_double)]


_lib = None


def _share_the_hip_runtime_with_torch():
    """PyTorch-ROCm wheels carry their own libamdhip64.so.  Two HIP runtimes in one process cannot both open the GPU: when this
    library (linked against /opt/rocm's runtime) is loaded BEFORE torch, a later ``torch.cuda`` call fails with "No HIP GPUs are
    available".  If torch is installed but not imported yet, its runtime is loaded here first, globally, so that this library and
    torch bind to the same one -- the state a process is in anyway when torch was imported first.  ``MPC_NO_TORCH_HIP=1`` skips it."""
    import sys
    if 'torch' in sys.modules or os.environ.get('MPC_NO_TORCH_HIP', '0') == '1':
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec('torch')
        if spec is None or not spec.origin:
            return
        path = os.path.join(os.path.dirname(spec.origin), 'lib', 'libamdhip64.so')
        if os.path.exists(path):
            ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    except Exception:       # no torch, or an unusual layout: the library's own runtime is used
        pass


def load():
    """Loads the HIP library; raises MpcError (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MpcError(f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                       f'(hipcc --offload-arch=gfx950). ppopt_amd has no CPU fallback.')
    _share_the_hip_runtime_with_torch()
    L = ctypes.CDLL(LIB_PATH)
    H = ctypes.c_void_p
    sig = {
        'mpc_device_count': (ctypes.c_int, []),
        'mpc_version': (ctypes.c_char_p, []),
        'mpc_last_global_error': (ctypes.c_char_p, []),
        'mpc_create': (ctypes.c_int, [ctypes.POINTER(MpcProblem), ctypes.c_int32, ctypes.c_void_p, ctypes.POINTER(H)]),
        'mpc_destroy': (ctypes.c_int, [H]),
        'mpc_last_error': (ctypes.c_char_p, [H]),
        'mpc_mask_words': (ctypes.c_int32, [H]),
        'mpc_set_region_overlap': (ctypes.c_int, [H, ctypes.c_int32]),
        'mpc_set_timing': (ctypes.c_int, [H, ctypes.c_int32]),
        'mpc_engine_kind': (ctypes.c_int, [H, ctypes.POINTER(ctypes.c_int32)]),
        'mpc_program_block': (ctypes.c_int, [H, ctypes.c_int32, _dp, ctypes.c_int64, _lp]),
        'mpc_region_doubles': (ctypes.c_int64, [H]),
        'mpc_region_ints': (ctypes.c_int64, [H]),
        'mpc_lds_bytes': (ctypes.c_int32, [H, ctypes.c_int32]),
        'mpc_stream': (ctypes.c_void_p, [H]),
        'mpc_frontier_root': (ctypes.c_int, [H]),
        'mpc_frontier_set': (ctypes.c_int, [H, _ip, ctypes.c_int64, ctypes.c_int32]),
        'mpc_frontier_set_device': (ctypes.c_int, [H, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32]),
        'mpc_frontier_info': (ctypes.c_int, [H, _lp, _ip]),
        'mpc_frontier_get': (ctypes.c_int, [H, _ip, ctypes.c_int64]),
        'mpc_pruned_clear': (ctypes.c_int, [H]),
        'mpc_pruned_add': (ctypes.c_int, [H, _u64p, ctypes.c_int64]),
        'mpc_pruned_add_device': (ctypes.c_int, [H, ctypes.c_void_p, ctypes.c_int64]),
        'mpc_pruned_count': (ctypes.c_int64, [H]),
        'mpc_pruned_get': (ctypes.c_int, [H, _u64p, ctypes.c_int64]),
        'mpc_level_run': (ctypes.c_int, [H, ctypes.c_int32, ctypes.POINTER(LevelStats)]),
        'mpc_level_run_ex': (ctypes.c_int, [H, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(LevelStats)]),
        'mpc_level_run_batch': (ctypes.c_int, [ctypes.POINTER(H), ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.c_int32, ctypes.POINTER(LevelStats), ctypes.POINTER(ctypes.c_int32)]),
        'mpc_level_batch_start': (ctypes.c_int, [ctypes.POINTER(H), ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.c_int32, ctypes.POINTER(ctypes.c_void_p)]),
        'mpc_level_batch_wait': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(LevelStats), ctypes.POINTER(ctypes.c_int32)]),
        'mpc_level_memory_gb': (ctypes.c_double, [H, ctypes.c_int32]),
        'mpc_frontier_advance_batch': (ctypes.c_int, [ctypes.POINTER(H), ctypes.c_int32]),
        'mpc_trim': (ctypes.c_int, [H]),
        'mpc_level_status': (ctypes.c_int, [H, _u8p]),
        'mpc_level_start': (ctypes.c_int, [H, ctypes.c_int32, ctypes.c_int32]),
        'mpc_level_stream_info': (ctypes.c_int, [H, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p),
                                                 ctypes.POINTER(ctypes.c_void_p), _lp, _lp, _ip, _ip]),
        'mpc_level_chunk_wait': (ctypes.c_int, [H, ctypes.c_int32]),
        'mpc_level_wait': (ctypes.c_int, [H, ctypes.POINTER(LevelStats)]),
        'mpc_level_stream_fixup': (ctypes.c_int, [H, _dp, _ip, _dp, _lp]),
        'mpc_base_result': (ctypes.c_int, [H, _u8p, _lp, _dp, _ip]),
        'mpc_solve_start': (ctypes.c_int, [H, ctypes.c_int32, ctypes.c_int32]),
        'mpc_solve_level': (ctypes.c_int, [H, ctypes.c_int32, ctypes.POINTER(SolveLevelInfo)]),
        'mpc_solve_chunk_wait': (ctypes.c_int, [H, ctypes.c_int32, ctypes.c_int32]),
        'mpc_solve_level_wait': (ctypes.c_int, [H, ctypes.c_int32, ctypes.POINTER(LevelStats), ctypes.POINTER(ctypes.c_double)]),
        'mpc_solve_wait': (ctypes.c_int, [H, ctypes.POINTER(ctypes.c_int32)]),
        'mpc_level_regions': (ctypes.c_int, [H, _dp, _ip, _lp, ctypes.c_int64]),
        'mpc_compact_strides': (ctypes.c_int, [H, _lp, _lp, _lp]),
        'mpc_level_regions_compact': (ctypes.c_int, [H, _dp, _ip, ctypes.c_int64, _dp, ctypes.c_int64, _lp, _lp]),
        'mpc_frontier_shard': (ctypes.c_int, [H, ctypes.c_int32, ctypes.c_int32]),
        'mpc_level_slots': (ctypes.c_int64, [H]),
        'mpc_level_regions_slots': (ctypes.c_int, [H, _dp, _ip, ctypes.c_int64, _dp, ctypes.c_int64, _lp, _lp]),
        'mpc_level_regions_slots_async': (ctypes.c_int, [H, _dp, _ip, ctypes.c_int64, _dp, ctypes.c_int64, _lp, _lp]),
        'mpc_level_regions_slots_nowait': (ctypes.c_int, [H, _dp, _ip, ctypes.c_int64, _dp, ctypes.c_int64, _lp, _lp]),
        'mpc_level_batch_fetch': (ctypes.c_int, [ctypes.POINTER(H), ctypes.c_int32, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), _lp, ctypes.POINTER(ctypes.c_void_p), _lp, _lp, _lp]),
        'mpc_sync': (ctypes.c_int, [H]),
        'mpc_fetch_wait': (ctypes.c_int, [ctypes.c_int32]),
        'mpc_solve_many_start': (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.c_int32, ctypes.POINTER(ctypes.c_void_p)]),
        'mpc_solve_many_level': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ManyLevelInfo)]),
        'mpc_solve_many_wait': (ctypes.c_int, [ctypes.c_void_p]),
        'mpc_locator_create': (ctypes.c_int, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, _lp, _dp, _dp, _dp, _dp, _dp,
                                               ctypes.POINTER(ctypes.c_void_p)]),
        'mpc_locator_query': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, _dp, ctypes.c_double, ctypes.c_int32, _lp, _dp,
                                              ctypes.POINTER(ctypes.c_float)]),
        'mpc_locator_destroy': (ctypes.c_int, [ctypes.c_void_p]),
        'mpc_locator_set_adjacency': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, _u64p, _ip]),
        'mpc_host_alloc': (ctypes.c_int, [ctypes.c_uint64, ctypes.POINTER(ctypes.c_void_p)]),
        'mpc_host_free': (ctypes.c_int, [ctypes.c_void_p]),
        'mpc_level_children': (ctypes.c_int, [H, _ip, ctypes.c_int64]),
        'mpc_level_children_device': (ctypes.c_int, [H, ctypes.c_void_p, ctypes.c_int64]),
        'mpc_level_pruned_new': (ctypes.c_int, [H, _u64p, ctypes.c_int64]),
        'mpc_level_pruned_new_device': (ctypes.c_int, [H, ctypes.c_void_p, ctypes.c_int64]),
        'mpc_level_regions_device': (ctypes.c_int, [H, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                                    ctypes.c_int64, _lp, _lp]),
        'mpc_frontier_advance': (ctypes.c_int, [H]),
        'mpc_qp_solve_batch': (ctypes.c_int, [H, ctypes.c_int64, _dp, _ip, _dp, _dp, _u8p, _ip]),
        'mpc_facet_centres': (ctypes.c_int, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, _lp, _dp, _dp, _dp, _ip]),
        'mpc_graph_begin': (ctypes.c_int, [H, _u64p, ctypes.c_int64, ctypes.c_int32]),
        'mpc_graph_wave': (ctypes.c_int, [H, _ip, _lp, ctypes.c_int32, _ip, _lp, _lp]),
        'mpc_graph_group_run': (ctypes.c_int, [H, ctypes.c_int32, ctypes.POINTER(LevelStats)]),
        'mpc_graph_wave_close': (ctypes.c_int, [H, _lp, _lp]),
        'mpc_check_level': (ctypes.c_int, [H, _ip, ctypes.c_int64, ctypes.c_int32, _u64p, ctypes.c_int64, ctypes.c_int32,
                                           _u8p, _lp, _dp, _ip, _lp, ctypes.c_int64, _lp, _ip, ctypes.c_int64]),
        'mpc_lp_solve_batch': (ctypes.c_int, [ctypes.c_int32, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, _dp,
                                              ctypes.c_int32, _dp, ctypes.c_int32, _dp, ctypes.c_int32, _u8p, _ip, _dp,
                                              _dp, _ip]),
    }
    for name, (res, args) in sig.items():
        if os.environ.get('MPC_LIB_ALLOW_MISSING') == '1' and not hasattr(L, name):
            continue      # tools/ab_lib.py comparing against an OLDER build of the library (MPC_LIB_PATH); never set in tests
        fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


EXPORTED_SYMBOLS = ['mpc_device_count', 'mpc_version', 'mpc_last_global_error', 'mpc_create', 'mpc_destroy',
                    'mpc_last_error', 'mpc_mask_words', 'mpc_set_region_overlap', 'mpc_set_timing', 'mpc_engine_kind', 'mpc_program_block', 'mpc_region_doubles', 'mpc_region_ints', 'mpc_lds_bytes', 'mpc_stream',
                    'mpc_frontier_root', 'mpc_frontier_set', 'mpc_frontier_set_device', 'mpc_frontier_info',
                    'mpc_frontier_get', 'mpc_pruned_clear', 'mpc_pruned_add', 'mpc_pruned_add_device',
                    'mpc_pruned_count', 'mpc_pruned_get', 'mpc_level_run', 'mpc_level_run_ex', 'mpc_level_run_batch', 'mpc_frontier_advance_batch', 'mpc_level_memory_gb', 'mpc_trim', 'mpc_level_batch_start', 'mpc_level_batch_wait', 'mpc_level_regions_slots_nowait', 'mpc_level_batch_fetch', 'mpc_level_status', 'mpc_level_start', 'mpc_level_stream_info', 'mpc_level_chunk_wait', 'mpc_level_wait', 'mpc_level_stream_fixup', 'mpc_base_result', 'mpc_solve_start', 'mpc_solve_level', 'mpc_solve_chunk_wait', 'mpc_solve_level_wait', 'mpc_solve_wait', 'mpc_level_regions', 'mpc_compact_strides',
                    'mpc_level_regions_compact', 'mpc_frontier_shard', 'mpc_level_slots', 'mpc_level_regions_slots', 'mpc_level_regions_slots_async', 'mpc_sync', 'mpc_fetch_wait', 'mpc_solve_many_start', 'mpc_solve_many_level', 'mpc_solve_many_wait', 'mpc_host_alloc', 'mpc_host_free', 'mpc_locator_create', 'mpc_locator_query', 'mpc_locator_destroy', 'mpc_locator_set_adjacency', 'mpc_level_children', 'mpc_level_children_device', 'mpc_level_pruned_new',
                    'mpc_level_pruned_new_device', 'mpc_level_regions_device', 'mpc_frontier_advance', 'mpc_qp_solve_batch', 'mpc_facet_centres', 'mpc_graph_begin', 'mpc_graph_wave', 'mpc_graph_group_run', 'mpc_graph_wave_close', 'mpc_check_level', 'mpc_lp_solve_batch']


def pinned_empty(shape, dtype) -> numpy.ndarray:
    """An uninitialised array in pooled page-locked host memory (mpc_host_alloc); the block goes back to the pool when
    the last view of the array is gone."""
    L = load()
    dtype = numpy.dtype(dtype)
    count = int(numpy.prod(shape))
    nbytes = max(count * dtype.itemsize, 1)
    ptr = ctypes.c_void_p()
    if L.mpc_host_alloc(nbytes, ctypes.byref(ptr)) != 0:
        raise MemoryError(f'mpc_host_alloc({nbytes}) failed')
    buf = (ctypes.c_char * nbytes).from_address(ptr.value)
    weakref.finalize(buf, L.mpc_host_free, ctypes.c_void_p(ptr.value))
    return numpy.frombuffer(buf, dtype=dtype, count=count).reshape(shape)


def pinned_adopt(ptr: int, shape, dtype) -> numpy.ndarray:
    """An array over a page-locked block the library has handed over (mpc_level_stream_info): same ownership rule as
    pinned_empty -- the block returns to the pool with the last view."""
    L = load()
    dtype = numpy.dtype(dtype)
    count = int(numpy.prod(shape))
    nbytes = max(count * dtype.itemsize, 1)
    buf = (ctypes.c_char * nbytes).from_address(ptr)
    weakref.finalize(buf, L.mpc_host_free, ctypes.c_void_p(ptr))
    return numpy.frombuffer(buf, dtype=dtype, count=count).reshape(shape)


def _f64(a):
    return numpy.ascontiguousarray(a, dtype=numpy.float64)


def sets_to_masks(sets, words: int = MASK_WORDS) -> numpy.ndarray:
    """Active sets (iterables of row indices < 64 * words) -> (m, words) uint64 bit masks (include/mpcombi.h,
    mpc_mask_words: 2 words for programs with n_c <= 128, 4 up to 256)."""
    out = numpy.zeros((len(sets), words), dtype=numpy.uint64)
    for i, s in enumerate(sets):
        for v in s:
            out[i, int(v) >> 6] |= numpy.uint64(1) << numpy.uint64(int(v) & 63)
    return out


def masks_to_sets(masks: numpy.ndarray, words: int = MASK_WORDS):
    out = []
    for row in numpy.asarray(masks, dtype=numpy.uint64).reshape(-1, words):
        s = []
        for w in range(words):
            v = int(row[w])
            while v:
                low = v & -v
                s.append(w * 64 + low.bit_length() - 1)
                v ^= low
        out.append(tuple(s))
    return out


class Engine:
    """One device-resident program (mpc_handle).  Mirrors what every reference worker holds: the presolved
    matrices of the program plus the pruned list; see include/mpcombi.h for the call-by-call mapping."""

    def __init__(self, A, b, F, c, H, Q, A_t, b_t, n_eq: int, device: int = 0, stream: Optional[int] = None):
        L = load()
        self._L = L
        self._h = ctypes.c_void_p()
        self.A, self.b, self.F = _f64(A), _f64(b).reshape(-1), _f64(F)
        self.c, self.H = _f64(c).reshape(-1), _f64(H)
        self.Q = None if Q is None else _f64(Q)
        self.A_t, self.b_t = _f64(A_t), _f64(b_t).reshape(-1)
        self.n_c, self.n_x = self.A.shape
        self.n_t = self.F.shape[1]
        self.n_tc = self.A_t.shape[0]
        self.n_eq = int(n_eq)
        if self.A_t.ndim != 2:
            self.A_t = self.A_t.reshape(self.n_tc, self.n_t)
        p = MpcProblem(self.n_x, self.n_t, self.n_c, self.n_eq, self.n_tc, self.A.ctypes.data_as(_dp),
                       self.b.ctypes.data_as(_dp), self.F.ctypes.data_as(_dp), self.c.ctypes.data_as(_dp),
                       self.H.ctypes.data_as(_dp), None if self.Q is None else self.Q.ctypes.data_as(_dp),
                       self.A_t.ctypes.data_as(_dp) if self.n_tc else None,
                       self.b_t.ctypes.data_as(_dp) if self.n_tc else None)
        rc = L.mpc_create(ctypes.byref(p), int(device), ctypes.c_void_p(stream) if stream else None,
                          ctypes.byref(self._h))
        if rc != MPC_OK:
            raise MpcError(f'mpc_create failed ({rc}): {L.mpc_last_global_error().decode()}')
        self.mask_words = int(L.mpc_mask_words(self._h))
        self.rec_d = int(L.mpc_region_doubles(self._h))
        self.rec_i = int(L.mpc_region_ints(self._h))
        self.device = int(device)
        self._twin = None

    # -- plumbing ----------------------------------------------------------------------------------------------
    def _check(self, rc, what):
        if rc != MPC_OK:
            cls = MpcCapacityError if rc == 3 and what in ('mpc_level_run', 'mpc_level_wait', 'mpc_level_run_batch', 'mpc_solve_wait') else MpcError
            raise cls(f'{what} failed ({rc}): {self._L.mpc_last_error(self._h).decode()}')

    def program_block(self, which: int) -> numpy.ndarray:
        """One of the program's one-off dense blocks as the device holds it (mpc_program_block): 0 W, 1 UV, 2 Gt, 3 X0H, 4 A A';
        5..10: the blocks with the equality rows eliminated (Wr, UVr, (A A')r, Me, Ne, gE; empty when not in use)."""
        n = ctypes.c_int64(0)
        ne = self.n_eq
        shape = {0: (self.n_c, self.n_c), 1: (self.n_c, self.n_t + 1), 2: (self.n_c, self.n_x), 3: (self.n_x, self.n_t + 1),
                 4: (self.n_c, self.n_c), 5: (self.n_c, self.n_c), 6: (self.n_c, self.n_t + 1), 7: (self.n_c, self.n_c),
                 8: (ne, self.n_t + 1), 9: (ne, self.n_c), 10: (2, ne)}[which]
        out = numpy.zeros(shape)
        self._check(self._L.mpc_program_block(self._h, which, out.ctypes.data_as(_dp), out.size, ctypes.byref(n)), 'mpc_program_block')
        return out if n.value else numpy.zeros((0, 0))

    def set_region_overlap(self, on: bool):
        self._check(self._L.mpc_set_region_overlap(self._h, 1 if on else 0), 'mpc_set_region_overlap')

    def set_timing(self, on: bool):
        """HIP-event records around the stages and heavy kernels of this handle's levels (mpc_set_timing): the drivers switch them on
        for the solves that ask for a profile; without them the ms_* fields of the level statistics are 0."""
        on = bool(on)
        if getattr(self, '_timing', None) is not on:
            self._check(self._L.mpc_set_timing(self._h, 1 if on else 0), 'mpc_set_timing')
            self._timing = on

    def engine_kind(self) -> dict:
        """Which kernels this handle's levels run on (mpc_engine_kind)."""
        out = (ctypes.c_int32 * 8)()
        self._check(self._L.mpc_engine_kind(self._h, out), 'mpc_engine_kind')
        return {'register_engine': bool(out[0]), 'rows_per_lane_theta': int(out[1]), 'rows_per_lane_x': int(out[2]), 'rows_per_lane_region': int(out[3]),
                'n_theta_instantiation': int(out[4]), 'theta_open': bool(out[5]), 'kkt_mode': int(out[6]), 'kkt_thread_max_rows': int(out[7])}

    def close(self):
        if getattr(self, '_twin', None) is not None:
            self._twin.close()
            self._twin = None
        if self._h:
            self._L.mpc_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stream(self) -> int:
        return int(self._L.mpc_stream(self._h) or 0)

    def lds_bytes(self, which: int = 0) -> int:
        return int(self._L.mpc_lds_bytes(self._h, which))

    # -- frontier / pruned list ----------------------------------------------------------------------------------
    def frontier_root(self):
        self._check(self._L.mpc_frontier_root(self._h), 'mpc_frontier_root')

    def frontier_set(self, cands: numpy.ndarray):
        cands = numpy.ascontiguousarray(cands, dtype=numpy.int32)
        n, k = cands.shape
        self._check(self._L.mpc_frontier_set(self._h, cands.ctypes.data_as(_ip), n, k), 'mpc_frontier_set')

    def frontier_set_device(self, ptr: int, n: int, k: int):
        self._check(self._L.mpc_frontier_set_device(self._h, ctypes.c_void_p(ptr), n, k), 'mpc_frontier_set_device')

    def frontier_info(self):
        n, k = ctypes.c_int64(0), ctypes.c_int32(0)
        self._check(self._L.mpc_frontier_info(self._h, ctypes.byref(n), ctypes.byref(k)), 'mpc_frontier_info')
        return n.value, k.value

    def frontier_get(self) -> numpy.ndarray:
        n, k = self.frontier_info()
        out = numpy.zeros((n, k), dtype=numpy.int32)
        if n:
            self._check(self._L.mpc_frontier_get(self._h, out.ctypes.data_as(_ip), n), 'mpc_frontier_get')
        return out

    def pruned_clear(self):
        self._check(self._L.mpc_pruned_clear(self._h), 'mpc_pruned_clear')

    def pruned_add(self, masks: numpy.ndarray):
        masks = numpy.ascontiguousarray(masks, dtype=numpy.uint64).reshape(-1, self.mask_words)
        if len(masks):
            self._check(self._L.mpc_pruned_add(self._h, masks.ctypes.data_as(_u64p), len(masks)), 'mpc_pruned_add')

    def pruned_add_device(self, ptr: int, m: int):
        self._check(self._L.mpc_pruned_add_device(self._h, ctypes.c_void_p(ptr), m), 'mpc_pruned_add_device')

    def frontier_shard(self, rank: int, world: int):
        self._check(self._L.mpc_frontier_shard(self._h, rank, world), 'mpc_frontier_shard')

    def pruned_count(self) -> int:
        return int(self._L.mpc_pruned_count(self._h))

    def pruned_get(self) -> numpy.ndarray:
        m = self.pruned_count()
        out = numpy.zeros((m, self.mask_words), dtype=numpy.uint64)
        if m:
            self._check(self._L.mpc_pruned_get(self._h, out.ctypes.data_as(_u64p), m), 'mpc_pruned_get')
        return out

    # -- one level --------------------------------------------------------------------------------------------------
    def level_run(self, gen_children: bool, graph: bool = False, keep_lowdim: bool = False, stream: bool = False) -> LevelStats:
        """One level over the resident frontier.  ``graph``: the question of the connected-graph traversal (MPC_LEVEL_GRAPH);
        ``keep_lowdim``: the serial driver's expansion rule (MPC_LEVEL_KEEP_LOWDIM); ``stream``: the region records are
        written into page-locked host arrays, complete on return, handed over by ``level_stream_info()``."""
        st = LevelStats()
        self._check(self._L.mpc_level_run_ex(self._h, int(bool(gen_children)), (MPC_LEVEL_GRAPH if graph else 0)
                                             | (MPC_LEVEL_KEEP_LOWDIM if keep_lowdim else 0) | (MPC_LEVEL_STREAM if stream else 0),
                                             ctypes.byref(st)), 'mpc_level_run')
        self._last = st
        return st

    @staticmethod
    def frontier_advance_batch(engines):
        """``frontier_advance`` for many engines with one call into the library (mpc_frontier_advance_batch)."""
        B = len(engines)
        if B:
            hs = (ctypes.c_void_p * B)(*[e._h for e in engines])
            rc = engines[0]._L.mpc_frontier_advance_batch(hs, B)
            if rc != 0:
                Engine._raise_for(engines, rc)

    def level_memory_gb(self, gen_children: bool) -> float:
        """Device memory (GB) the next level of this engine's frontier holds in a batch (mpc_level_memory_gb)."""
        return float(self._L.mpc_level_memory_gb(self._h, int(bool(gen_children))))

    def trim(self):
        """Gives the level buffers back to the pool (mpc_trim); the program stays usable."""
        self._check(self._L.mpc_trim(self._h), 'mpc_trim')

    @staticmethod
    def level_batch_start(engines, gen_children, keep_lowdim: bool = False):
        """First half of ``level_run_batch`` (mpc_level_batch_start): queues the shared launches of one level for all engines and
        returns a token for ``level_batch_wait``; the engines must not be touched in between."""
        B = len(engines)
        L = engines[0]._L
        hs = (ctypes.c_void_p * B)(*[e._h for e in engines])
        gc = (ctypes.c_int32 * B)(*[int(bool(g)) for g in gen_children])
        tok = ctypes.c_void_p()
        rc = L.mpc_level_batch_start(hs, B, gc, MPC_LEVEL_KEEP_LOWDIM if keep_lowdim else 0, ctypes.byref(tok))
        if rc != 0:
            Engine._raise_for(engines, rc)
        return (tok, list(engines))

    @staticmethod
    def level_batch_wait(token):
        """Second half: ([LevelStats per engine], members that went through the shared launches)."""
        tok, engines = token
        B = len(engines)
        stats = (LevelStats * B)()
        nb = ctypes.c_int32(0)
        rc = engines[0]._L.mpc_level_batch_wait(tok, stats, ctypes.byref(nb))
        if rc != 0:
            Engine._raise_for(engines, rc)
        out = list(stats)               # elements are views that keep the array alive
        for e, st in zip(engines, out):
            e._last = st
        return out, int(nb.value)

    @staticmethod
    def level_batch_fetch(engines):
        """``level_regions_slots_nowait`` for many engines with ONE call into the library and THREE page-locked blocks for all of
        them (the members' arrays are views into those): [(head_d, head_i, erows, k) per engine], copies queued only.  Sizes come from
        each engine's LevelStats of the level just run."""
        B = len(engines)
        if B == 0:
            return []
        fds, fis, nss, nrs = [], [], [], []
        for e in engines:
            st = e._last
            k = int(st.k)
            nr = int(st.n_regions)
            fds.append(e.n_x * e.n_t + e.n_x + k * e.n_t + k)
            fis.append(8 + k + e.n_tc + k + 2 * (e.n_c - k))
            nss.append(int(st.n_opt) if nr else 0)
            rows_t = e.n_c - e.n_eq + e.n_tc
            # mpc_compact_strides' max_rows: pooled rows + a full row block per candidate the LDS-engine kernel re-solved; a level
            # built by that kernel alone (no register-resident instantiation: one parameter) reports no pooled rows
            nrs.append(0 if not nr else (int(st.n_region_rows) + int(st.n_region_retry) * rows_t if st.n_region_rows else nr * rows_t))
        od = numpy.concatenate([[0], numpy.cumsum([n * f for n, f in zip(nss, fds)])]).astype(numpy.int64)
        oi = numpy.concatenate([[0], numpy.cumsum([n * f for n, f in zip(nss, fis)])]).astype(numpy.int64)
        oe = numpy.concatenate([[0], numpy.cumsum([n * (e.n_t + 1) for n, e in zip(nrs, engines)])]).astype(numpy.int64)
        big_d = pinned_empty((max(int(od[-1]), 1),), numpy.float64)
        big_i = pinned_empty((max(int(oi[-1]), 1),), numpy.int32)
        big_e = pinned_empty((max(int(oe[-1]), 1),), numpy.float64)
        pd = (ctypes.c_void_p * B)(*(big_d.ctypes.data + 8 * od[:-1]).tolist())
        pi = (ctypes.c_void_p * B)(*(big_i.ctypes.data + 4 * oi[:-1]).tolist())
        pe = (ctypes.c_void_p * B)(*(big_e.ctypes.data + 8 * oe[:-1]).tolist())
        caps = (ctypes.c_int64 * B)(*nss)
        capr = (ctypes.c_int64 * B)(*nrs)
        n1 = (ctypes.c_int64 * B)()
        n2 = (ctypes.c_int64 * B)()
        hs = (ctypes.c_void_p * B)(*[e._h for e in engines])
        rc = engines[0]._L.mpc_level_batch_fetch(hs, B, pd, pi, caps, pe, capr, n1, n2)
        if rc != 0:
            Engine._raise_for(engines, rc)
        # (the members' arrays are complete after Engine.fetch_wait / any member's sync / the next level_batch_start)
        out = []
        for j, e in enumerate(engines):
            ns, nr = int(n1[j]), int(n2[j])
            out.append((big_d[od[j]:od[j] + ns * fds[j]].reshape(ns, fds[j]), big_i[oi[j]:oi[j] + ns * fis[j]].reshape(ns, fis[j]),
                        big_e[oe[j]:oe[j] + nr * (e.n_t + 1)].reshape(nr, e.n_t + 1), int(e._last.k)))
        return out

    @staticmethod
    def solve_many_start(engines, max_levels, keep_lowdim: bool = False, base: bool = False):
        """The level loop of all ``engines`` on a thread of the library (mpc_solve_many_start): every engine's pruned list is cleared, its
        frontier rooted, then level after level with shared launches; returns the job for ``solve_many_level`` / ``solve_many_wait``.
        The engines must not be touched until ``solve_many_wait`` has returned."""
        B = len(engines)
        hs = (ctypes.c_void_p * B)(*[e._h for e in engines])
        ml = (ctypes.c_int32 * B)(*[int(v) for v in max_levels])
        job = ctypes.c_void_p()
        rc = engines[0]._L.mpc_solve_many_start(hs, B, ml, (MPC_LEVEL_KEEP_LOWDIM if keep_lowdim else 0) | (MPC_SOLVE_MANY_BASE if base else 0), ctypes.byref(job))
        if rc != 0:
            Engine._raise_for(engines, rc)
        return (job, list(engines))

    @staticmethod
    def solve_many_level(job, level: int):
        """Blocks until level ``level`` of the job has been run and its records are complete (mpc_solve_many_level).  Returns
        ``(done, None)`` past the last level (done = 1: all members finished, 2: the caller takes over -- memory budget), else
        ``(0, (members, stats, n_shared, ms_wall, records, base))`` (``base``: the closing level of the base active sets) with ``records`` = [(member, head_d, head_i, erows, k)] for the members
        that found regions (views into three page-locked blocks that return to the pool with their last view)."""
        jb, engines = job
        info = ManyLevelInfo()
        rc = engines[0]._L.mpc_solve_many_level(jb, int(level), ctypes.byref(info))
        if rc != 0:
            Engine._raise_for(engines, rc)
        nm = int(info.n_members)
        if nm == 0:
            return int(info.done), None
        members = info.member[:nm]
        # copies: info.stats[j] is a view into the job's own vector, which mpc_solve_many_wait frees (the engines keep `_last`)
        stats = [LevelStats.from_buffer_copy(info.stats[j]) for j in range(nm)]
        for i, st in zip(members, stats):
            engines[i]._last = st
        records = []
        if info.head_d:
            big_d = pinned_adopt(info.head_d, (int(info.len_d),), numpy.float64)
            big_i = pinned_adopt(info.head_i, (int(info.len_i),), numpy.int32)
            big_e = pinned_adopt(info.erows, (max(int(info.len_e), 1),), numpy.float64)
            ns, nr, od, oi, oe = info.n_slots[:nm], info.n_rows[:nm], info.off_d[:nm], info.off_i[:nm], info.off_e[:nm]
            for j, i in enumerate(members):
                if ns[j] == 0:
                    continue
                e, k = engines[i], int(stats[j].k)
                fd = e.n_x * e.n_t + e.n_x + k * e.n_t + k
                fi = 8 + k + e.n_tc + k + 2 * (e.n_c - k)
                records.append((i, big_d[od[j]:od[j] + ns[j] * fd].reshape(ns[j], fd), big_i[oi[j]:oi[j] + ns[j] * fi].reshape(ns[j], fi),
                                big_e[oe[j]:oe[j] + nr[j] * (e.n_t + 1)].reshape(nr[j], e.n_t + 1), k))
        return 0, (members, stats, int(info.n_shared), float(info.ms_wall), records, bool(info.base))

    @staticmethod
    def solve_many_wait(job):
        """Joins the loop and frees the job (mpc_solve_many_wait); raises what the loop failed with."""
        jb, engines = job
        rc = engines[0]._L.mpc_solve_many_wait(jb)
        if rc != 0:
            Engine._raise_for(engines, rc)

    @staticmethod
    def fetch_wait(engine):
        """Waits for the shared record copy of the last ``level_batch_fetch`` on the engine's device (mpc_fetch_wait)."""
        rc = engine._L.mpc_fetch_wait(int(engine.device))
        if rc != 0:
            engine._check(rc, 'mpc_fetch_wait')

    @staticmethod
    def _raise_for(engines, rc):
        for e in engines:      # the failing member carries the message
            if e._L.mpc_last_error(e._h):
                e._check(rc, 'mpc_level_run_batch')
        engines[0]._check(rc, 'mpc_level_run_batch')

    @staticmethod
    def level_run_batch(engines, gen_children, keep_lowdim: bool = False):
        """One level of several programs per launch (mpc_level_run_batch): ``engines`` are distinct Engines on one device, each
        with its frontier; ``gen_children`` one flag per engine.  Returns ([LevelStats per engine], members that went through the
        shared launches).  Afterwards every engine is in the state ``level_run`` would have left it in."""
        return Engine.level_batch_wait(Engine.level_batch_start(engines, gen_children, keep_lowdim))

    # -- the same level on the handle's worker thread, region records streamed to the host (include/mpcombi.h) ----------
    def level_start(self, gen_children: bool, stream: bool = True, then_base: bool = False, keep_lowdim: bool = False,
                    only_base: bool = False):
        self._check(self._L.mpc_level_start(self._h, int(bool(gen_children)),
                                            (MPC_LEVEL_STREAM if stream else 0) | (MPC_LEVEL_THEN_BASE if then_base else 0)
                                            | (MPC_LEVEL_KEEP_LOWDIM if keep_lowdim else 0)
                                            | (MPC_LEVEL_ONLY_BASE if only_base else 0)), 'mpc_level_start')

    def twin(self) -> 'Engine':
        """A second handle of the same program on the same device (created on first use, ~0.5 ms): the solve loop runs the
        base-set check there (``level_start(False, only_base=True)``) while this handle works on its large levels."""
        if self._twin is None:
            self._twin = Engine(self.A, self.b, self.F, self.c, self.H, self.Q, self.A_t, self.b_t, self.n_eq, device=self.device)
        return self._twin

    def base_result(self):
        """(status, rec_d [1, rec_d] or empty, rec_i) of the base-set check the worker ran behind the last level
        (``level_start(..., then_base=True)``), or None when it did not run."""
        status = numpy.zeros(1, dtype=numpy.uint8)
        nreg = ctypes.c_int64(0)
        d = numpy.zeros((1, self.rec_d))
        i = numpy.zeros((1, self.rec_i), dtype=numpy.int32)
        rc = self._L.mpc_base_result(self._h, status.ctypes.data_as(_u8p), ctypes.byref(nreg), d.ctypes.data_as(_dp), i.ctypes.data_as(_ip))
        if rc != MPC_OK:
            return None
        return status, d[:nreg.value], i[:nreg.value]

    def level_stream_info(self):
        """Blocks until the running level's region stage has been launched.  None when the level does not stream, else
        (head_d [S, fd], head_i [S, fi], erows [cap_rows, n_t+1], chunk, n_chunks) -- arrays the region kernel is writing;
        chunk j (slots j*chunk ...) may be read after ``level_chunk_wait(j)``."""
        hd, hi, er = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        ns, cr, ch, nch = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int32(0), ctypes.c_int32(0)
        self._check(self._L.mpc_level_stream_info(self._h, ctypes.byref(hd), ctypes.byref(hi), ctypes.byref(er), ctypes.byref(ns),
                                                  ctypes.byref(cr), ctypes.byref(ch), ctypes.byref(nch)), 'mpc_level_stream_info')
        if ns.value == 0:
            return None
        n, k = self.frontier_info()
        fd = self.n_x * self.n_t + self.n_x + k * self.n_t + k
        fi = 8 + k + self.n_tc + k + 2 * (self.n_c - k)
        return (pinned_adopt(hd.value, (ns.value, fd), numpy.float64), pinned_adopt(hi.value, (ns.value, fi), numpy.int32),
                pinned_adopt(er.value, (max(cr.value, 1), self.n_t + 1), numpy.float64), int(ch.value), int(nch.value))

    def level_chunk_wait(self, j: int):
        self._check(self._L.mpc_level_chunk_wait(self._h, int(j)), 'mpc_level_chunk_wait')

    def level_wait(self) -> LevelStats:
        st = LevelStats()
        self._check(self._L.mpc_level_wait(self._h, ctypes.byref(st)), 'mpc_level_wait')
        self._last = st
        return st

    def level_stream_fixup(self, hd, hi, er) -> int:
        """Fills the slots of candidates the LDS-engine kernel re-solved (stats.n_region_retry > 0); returns the rows in use."""
        nrows = ctypes.c_int64(0)
        self._check(self._L.mpc_level_stream_fixup(self._h, hd.ctypes.data_as(_dp), hi.ctypes.data_as(_ip), er.ctypes.data_as(_dp),
                                                   ctypes.byref(nrows)), 'mpc_level_stream_fixup')
        return int(nrows.value)

    # -- the whole level loop on the handle's worker thread (include/mpcombi.h, mpc_solve_*) ---------------------------------
    def solve_start(self, max_levels: int, stream: bool = True, fetch: bool = True, then_base: bool = False, keep_lowdim: bool = False):
        self._check(self._L.mpc_solve_start(self._h, int(max_levels), (MPC_LEVEL_STREAM if stream else 0) | (MPC_SOLVE_FETCH if fetch else 0)
                                            | (MPC_LEVEL_THEN_BASE if then_base else 0) | (MPC_LEVEL_KEEP_LOWDIM if keep_lowdim else 0)), 'mpc_solve_start')

    def solve_level(self, level: int):
        """Blocks until level ``level`` of the running solve has records to hand over or has finished.  None: the loop ended before
        this level.  Else (mode, k, n, head_d, head_i, erows, chunk, n_chunks): mode 1 = arrays the region kernel is still writing
        (chunk j readable after ``solve_chunk_wait(level, j)``), 2 = complete arrays, 0 = no records (arrays None)."""
        info = SolveLevelInfo()
        rc = self._L.mpc_solve_level(self._h, int(level), ctypes.byref(info))
        if info.mode < 0:
            return None
        self._check(rc, 'mpc_solve_level')
        if info.mode == 0:
            return 0, int(info.k), int(info.n), None, None, None, 0, 0
        k = int(info.k)
        fd = self.n_x * self.n_t + self.n_x + k * self.n_t + k
        fi = 8 + k + self.n_tc + k + 2 * (self.n_c - k)
        ns = int(info.n_slots)
        return (int(info.mode), k, int(info.n), pinned_adopt(info.head_d, (ns, fd), numpy.float64), pinned_adopt(info.head_i, (ns, fi), numpy.int32),
                pinned_adopt(info.erows, (max(int(info.n_rows), 1), self.n_t + 1), numpy.float64), int(info.chunk), int(info.n_chunks))

    def solve_chunk_wait(self, level: int, j: int):
        self._check(self._L.mpc_solve_chunk_wait(self._h, int(level), int(j)), 'mpc_solve_chunk_wait')

    def solve_level_wait(self, level: int):
        """(LevelStats, wall milliseconds) of a finished level of the running solve."""
        st = LevelStats()
        ms = ctypes.c_double(0.0)
        self._check(self._L.mpc_solve_level_wait(self._h, int(level), ctypes.byref(st), ctypes.byref(ms)), 'mpc_solve_level_wait')
        self._last = st
        return st, float(ms.value)

    def solve_wait(self) -> int:
        """Joins the solve loop; raises what the loop failed with (MpcCapacityError as for ``level_wait``); levels completed."""
        nl = ctypes.c_int32(0)
        self._check(self._L.mpc_solve_wait(self._h, ctypes.byref(nl)), 'mpc_solve_wait')
        return int(nl.value)

    def qp_solve_batch(self, thetas: numpy.ndarray):
        """The program's QP at every row of ``thetas`` [m, n_t] (positive definite Q), one wavefront per point:
        (status [m] (0 optimal, 1 infeasible, 3 iteration limit), x [m, n_x], lambda [m, n_c], active [m, n_c] bool)."""
        th = _f64(thetas).reshape(-1, self.n_t)
        m = len(th)
        status = numpy.zeros(m, dtype=numpy.int32)
        x = numpy.zeros((m, self.n_x))
        lam = numpy.zeros((m, self.n_c))
        act = numpy.zeros((m, self.n_c), dtype=numpy.uint8)
        self._check(self._L.mpc_qp_solve_batch(self._h, m, th.ctypes.data_as(_dp), status.ctypes.data_as(_ip), x.ctypes.data_as(_dp),
                                               lam.ctypes.data_as(_dp), act.ctypes.data_as(_u8p), None), 'mpc_qp_solve_batch')
        return status, x, lam, act.astype(bool)

    # -- connected-graph traversal with the bookkeeping on the device (include/mpcombi.h, mpc_graph_*) -------------------
    def graph_begin(self, seed_masks: numpy.ndarray, variant: int):
        m = numpy.ascontiguousarray(seed_masks, dtype=numpy.uint64).reshape(-1, self.mask_words)
        self._check(self._L.mpc_graph_begin(self._h, m.ctypes.data_as(_u64p), len(m), int(variant)), 'mpc_graph_begin')

    def graph_wave(self):
        """[(cardinality, number of active sets)] of the current wave (empty: the traversal is complete), sets queued so far."""
        cap = 600
        ks = numpy.zeros(cap, dtype=numpy.int32)
        cs = numpy.zeros(cap, dtype=numpy.int64)
        ng = ctypes.c_int32(0)
        nw, nv = ctypes.c_int64(0), ctypes.c_int64(0)
        self._check(self._L.mpc_graph_wave(self._h, ks.ctypes.data_as(_ip), cs.ctypes.data_as(_lp), cap, ctypes.byref(ng), ctypes.byref(nw),
                                           ctypes.byref(nv)), 'mpc_graph_wave')
        return list(zip(ks[:ng.value].tolist(), cs[:ng.value].tolist())), int(nv.value)

    def graph_group_run(self, group: int) -> LevelStats:
        st = LevelStats()
        self._check(self._L.mpc_graph_group_run(self._h, int(group), ctypes.byref(st)), 'mpc_graph_group_run')
        self._last = st
        return st

    def graph_wave_close(self):
        nn, nv = ctypes.c_int64(0), ctypes.c_int64(0)
        self._check(self._L.mpc_graph_wave_close(self._h, ctypes.byref(nn), ctypes.byref(nv)), 'mpc_graph_wave_close')
        return int(nn.value), int(nv.value)

    def level_status(self) -> numpy.ndarray:
        n, _ = self.frontier_info()
        out = numpy.zeros(n, dtype=numpy.uint8)
        if n:
            self._check(self._L.mpc_level_status(self._h, out.ctypes.data_as(_u8p)), 'mpc_level_status')
        return out

    def level_regions(self):
        nr = int(self._last.n_regions)
        d = numpy.zeros((nr, self.rec_d))
        i = numpy.zeros((nr, self.rec_i), dtype=numpy.int32)
        idx = numpy.zeros(nr, dtype=numpy.int64)
        if nr:
            self._check(self._L.mpc_level_regions(self._h, d.ctypes.data_as(_dp), i.ctypes.data_as(_ip),
                                                  idx.ctypes.data_as(_lp), nr), 'mpc_level_regions')
        return d, i, idx

    def level_regions_compact(self):
        """Regions of the level in the device's compact form: (head_d [n, fd], head_i [n, fi], erows [R, n_t+1], k)."""
        nr = int(self._last.n_regions)
        fd, fi, mr = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
        self._check(self._L.mpc_compact_strides(self._h, ctypes.byref(fd), ctypes.byref(fi), ctypes.byref(mr)),
                    'mpc_compact_strides')
        k = int(self._last.k)
        rows_cap = int(mr.value) if nr else 0
        hd = numpy.empty((nr, fd.value))
        hi = numpy.empty((nr, fi.value), dtype=numpy.int32)
        er = numpy.empty((max(rows_cap, 1), self.n_t + 1))
        n1, n2 = ctypes.c_int64(0), ctypes.c_int64(0)
        if nr:
            self._check(self._L.mpc_level_regions_compact(self._h, hd.ctypes.data_as(_dp), hi.ctypes.data_as(_ip), nr,
                                                          er.ctypes.data_as(_dp), rows_cap, ctypes.byref(n1),
                                                          ctypes.byref(n2)), 'mpc_level_regions_compact')
        return hd[:n1.value], hi[:n1.value], er[:n2.value], k

    def sync(self):
        """Waits for everything queued on the handle's stream (completes an asynchronous slot fetch)."""
        self._check(self._L.mpc_sync(self._h), 'mpc_sync')

    def level_regions_slots_nowait(self):
        """``level_regions_slots`` that only QUEUES the copies: (head_d, head_i, erows, k) -- all three complete after ``sync()`` or
        the next level of this engine; then ``numpy.flatnonzero(head_i[:, 0] == REGION)`` are the region slots."""
        nr = int(self._last.n_regions)
        fd, fi, mr = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
        self._check(self._L.mpc_compact_strides(self._h, ctypes.byref(fd), ctypes.byref(fi), ctypes.byref(mr)), 'mpc_compact_strides')
        ns = int(self._L.mpc_level_slots(self._h)) if nr else 0
        rows_cap = int(mr.value) if nr else 0
        hd = pinned_empty((ns, fd.value), numpy.float64)
        hi = pinned_empty((ns, fi.value), numpy.int32)
        er = pinned_empty((max(rows_cap, 1), self.n_t + 1), numpy.float64)
        n1, n2 = ctypes.c_int64(0), ctypes.c_int64(0)
        if nr:
            self._check(self._L.mpc_level_regions_slots_nowait(self._h, hd.ctypes.data_as(_dp), hi.ctypes.data_as(_ip), ns, er.ctypes.data_as(_dp),
                                                               rows_cap, ctypes.byref(n1), ctypes.byref(n2)), 'mpc_level_regions_slots')
        return hd[:n1.value], hi[:n1.value], er[:n2.value], int(self._last.k)

    def level_regions_slots(self, early_return: bool = False):
        """All slots the region kernel wrote for this level, copied by DMA into pooled page-locked arrays:
        (head_d [S, fd], head_i [S, fi], erows [R, n_t+1], k, region_slots) -- region_slots = the slots that are regions.
        ``early_return``: come back when head_i is there; head_d and erows are complete after ``sync()`` (or the next
        ``level_run``) -- the caller must not read them before."""
        nr = int(self._last.n_regions)
        fd, fi, mr = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
        self._check(self._L.mpc_compact_strides(self._h, ctypes.byref(fd), ctypes.byref(fi), ctypes.byref(mr)),
                    'mpc_compact_strides')
        k = int(self._last.k)
        ns = int(self._L.mpc_level_slots(self._h)) if nr else 0
        rows_cap = int(mr.value) if nr else 0
        hd = pinned_empty((ns, fd.value), numpy.float64)
        hi = pinned_empty((ns, fi.value), numpy.int32)
        er = pinned_empty((max(rows_cap, 1), self.n_t + 1), numpy.float64)
        n1, n2 = ctypes.c_int64(0), ctypes.c_int64(0)
        if nr:
            fetch = self._L.mpc_level_regions_slots_async if early_return else self._L.mpc_level_regions_slots
            self._check(fetch(self._h, hd.ctypes.data_as(_dp), hi.ctypes.data_as(_ip), ns, er.ctypes.data_as(_dp), rows_cap,
                              ctypes.byref(n1), ctypes.byref(n2)), 'mpc_level_regions_slots')
        hi = hi[:n1.value]
        return hd[:n1.value], hi, er[:n2.value], k, numpy.flatnonzero(hi[:, 0] == REGION)

    def level_children(self) -> numpy.ndarray:
        n = int(self._last.n_children)
        out = numpy.zeros((n, int(self._last.k) + 1), dtype=numpy.int32)
        if n:
            self._check(self._L.mpc_level_children(self._h, out.ctypes.data_as(_ip), n), 'mpc_level_children')
        return out

    def level_children_device(self, ptr: int, cap: int):
        self._check(self._L.mpc_level_children_device(self._h, ctypes.c_void_p(ptr), cap), 'mpc_level_children_device')

    def level_pruned_new(self) -> numpy.ndarray:
        m = int(self._last.n_pruned_new)
        out = numpy.zeros((m, self.mask_words), dtype=numpy.uint64)
        if m:
            self._check(self._L.mpc_level_pruned_new(self._h, out.ctypes.data_as(_u64p), m), 'mpc_level_pruned_new')
        return out

    def level_region_shapes(self):
        """(slots, fd, fi, row capacity) of the level just run: the shapes level_regions_device fills."""
        fd, fi, mr = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
        self._check(self._L.mpc_compact_strides(self._h, ctypes.byref(fd), ctypes.byref(fi), ctypes.byref(mr)),
                    'mpc_compact_strides')
        ns = int(self._L.mpc_level_slots(self._h)) if int(self._last.n_regions) else 0
        return ns, int(fd.value), int(fi.value), int(mr.value) if ns else 0

    def level_regions_device(self, hd_ptr: int, hi_ptr: int, er_ptr: int, cap_slots: int, cap_rows: int):
        """Slot arrays device -> device into caller-owned device buffers; (n_slots, n_rows), or None when some record
        of the level only exists in the host route's form (then use level_regions_slots)."""
        n1, n2 = ctypes.c_int64(0), ctypes.c_int64(0)
        rc = self._L.mpc_level_regions_device(self._h, ctypes.c_void_p(hd_ptr), ctypes.c_void_p(hi_ptr),
                                              ctypes.c_void_p(er_ptr), cap_slots, cap_rows, ctypes.byref(n1), ctypes.byref(n2))
        if rc == 4:      # MPC_ERR_STATE
            return None
        self._check(rc, 'mpc_level_regions_device')
        return int(n1.value), int(n2.value)

    def level_pruned_new_device(self, ptr: int, cap: int):
        self._check(self._L.mpc_level_pruned_new_device(self._h, ctypes.c_void_p(ptr), cap),
                    'mpc_level_pruned_new_device')

    def frontier_advance(self):
        self._check(self._L.mpc_frontier_advance(self._h), 'mpc_frontier_advance')

    # -- the host-buffer operator (pool.map(full_process) replacement) --------------------------------------------
    def check_level(self, cands: numpy.ndarray, pruned_masks: numpy.ndarray, gen_children: bool):
        cands = numpy.ascontiguousarray(cands, dtype=numpy.int32)
        n, k = cands.shape
        pm = numpy.ascontiguousarray(pruned_masks, dtype=numpy.uint64).reshape(-1, self.mask_words)
        status = numpy.zeros(n, dtype=numpy.uint8)
        nreg, nch = ctypes.c_int64(0), ctypes.c_int64(0)
        # small levels: capacities that cannot be exceeded, so one call suffices; large levels: ask first
        rcap = n if n <= 4096 else 0
        ccap = n * self.n_c if (gen_children and n <= 4096) else 0
        for _ in range(2):
            d = numpy.zeros((max(rcap, 1), self.rec_d))
            i = numpy.zeros((max(rcap, 1), self.rec_i), dtype=numpy.int32)
            idx = numpy.zeros(max(rcap, 1), dtype=numpy.int64)
            ch = numpy.zeros((max(ccap, 1), k + 1), dtype=numpy.int32)
            rc = self._L.mpc_check_level(self._h, cands.ctypes.data_as(_ip), n, k, pm.ctypes.data_as(_u64p), len(pm),
                                         int(bool(gen_children)), status.ctypes.data_as(_u8p), ctypes.byref(nreg),
                                         d.ctypes.data_as(_dp), i.ctypes.data_as(_ip), idx.ctypes.data_as(_lp), rcap,
                                         ctypes.byref(nch), ch.ctypes.data_as(_ip), ccap)
            if rc == MPC_ERR_CAPACITY:
                rcap, ccap = nreg.value, nch.value
                continue
            self._check(rc, 'mpc_check_level')
            break
        return status, d[:nreg.value], i[:nreg.value], idx[:nreg.value], ch[:nch.value]


def lp_solve_batch(A, b, c, eq_flags, device: int = 0, want_x: bool = True):
    """Batched LPs on the device.  A: (n_lp, m, n) or (m, n) shared; b likewise; c (n_lp, n) / (n,) / None;
    eq_flags: (n_lp, m) bool.  Returns (status, x, obj, iterations); x is None with ``want_x=False`` (nothing is
    allocated or copied back for it)."""
    L = load()
    A = _f64(A)
    eq_flags = numpy.ascontiguousarray(eq_flags, dtype=numpy.uint8)
    n_lp, m = eq_flags.shape
    shared_A = A.ndim == 2
    n = A.shape[-1]
    b = _f64(b)
    shared_b = b.size == m and n_lp != 1 or b.size == m
    if b.size != m:
        shared_b = False
    cp, shared_c = None, 1
    if c is not None:
        c = _f64(c)
        shared_c = int(c.size == n)
        cp = c.ctypes.data_as(_dp)
    status = numpy.zeros(n_lp, dtype=numpy.int32)
    x = numpy.zeros((n_lp, n)) if want_x else None
    obj = numpy.zeros(n_lp)
    it = numpy.zeros(n_lp, dtype=numpy.int32)
    rc = L.mpc_lp_solve_batch(int(device), n_lp, m, n, A.ctypes.data_as(_dp), int(shared_A), b.ctypes.data_as(_dp),
                              int(shared_b), cp, shared_c, eq_flags.ctypes.data_as(_u8p), status.ctypes.data_as(_ip),
                              None if x is None else x.ctypes.data_as(_dp), obj.ctypes.data_as(_dp), it.ctypes.data_as(_ip))
    if rc != MPC_OK:
        raise MpcError(f'mpc_lp_solve_batch failed ({rc}): {L.mpc_last_global_error().decode()}')
    return status, x, obj, it


def facet_centres(ef_rows: numpy.ndarray, row_off: numpy.ndarray, device: int = 0):
    """Chebyshev centre [R, n_t], radius [R] and LP status [R] of every facet (row) of the stacked polytopes
    (include/mpcombi.h, mpc_facet_centres)."""
    L = load()
    ef = _f64(ef_rows)
    off = numpy.ascontiguousarray(row_off, dtype=numpy.int64)
    R, n_t = len(ef), ef.shape[1] - 1
    centre, radius, status = numpy.zeros((R, n_t)), numpy.zeros(R), numpy.zeros(R, dtype=numpy.int32)
    rc = L.mpc_facet_centres(int(device), n_t, len(off) - 1, off.ctypes.data_as(_lp), ef.ctypes.data_as(_dp), centre.ctypes.data_as(_dp),
                             radius.ctypes.data_as(_dp), status.ctypes.data_as(_ip))
    if rc != MPC_OK:
        raise MpcError(f'mpc_facet_centres failed ({rc}): {L.mpc_last_global_error().decode()}')
    return centre, radius, status


class Locator:
    """Batched point location over stacked critical regions (include/mpcombi.h, mpc_locator_*).
    ef_rows: [total_rows, n_t+1] = [f | E] stacked; row_off: [n_regions+1]; xlaw: [n_regions, n_x, n_t+1] = [b | A]."""

    def __init__(self, row_off, ef_rows, xlaw, Q=None, c=None, H=None, device: int = 0):
        self._L = load()
        self.row_off = numpy.ascontiguousarray(row_off, dtype=numpy.int64)
        self.ef = _f64(ef_rows)
        self.xlaw = _f64(xlaw)
        self.n_regions = len(self.row_off) - 1
        self.n_x, self.n_t = int(self.xlaw.shape[1]), int(self.xlaw.shape[2]) - 1
        opt = [None if a is None else _f64(a) for a in (Q, c, H)]
        self._keep = opt
        ptr = ctypes.c_void_p()
        rc = self._L.mpc_locator_create(device, self.n_x, self.n_t, self.n_regions, self.row_off.ctypes.data_as(_lp),
                                        self.ef.ctypes.data_as(_dp), self.xlaw.ctypes.data_as(_dp),
                                        *[None if a is None else a.ctypes.data_as(_dp) for a in opt], ctypes.byref(ptr))
        if rc != MPC_OK:
            raise MpcError(f'mpc_locator_create failed ({rc}): {self._L.mpc_last_global_error().decode()}')
        self._h = ptr
        self.last_ms = 0.0
        self.has_adjacency = False

    def set_adjacency(self, masks: numpy.ndarray, row_info: numpy.ndarray, n_c: int) -> bool:
        """Facet adjacency for the walk (mpc_locator_set_adjacency): masks [n_regions, words] uint64, row_info [total_rows]
        int32 = kind << 16 | id.  False when the library refuses it (two regions with one active set)."""
        m = numpy.ascontiguousarray(masks, dtype=numpy.uint64)
        ri = numpy.ascontiguousarray(row_info, dtype=numpy.int32)
        rc = self._L.mpc_locator_set_adjacency(self._h, int(m.shape[1]), int(n_c), m.ctypes.data_as(_u64p), ri.ctypes.data_as(_ip))
        self.has_adjacency = rc == MPC_OK
        return self.has_adjacency

    def query(self, theta: numpy.ndarray, tol: float = 1e-5, overlapping: bool = False, want_x: bool = True,
              inclusive: bool = False, walk: bool = False):
        """theta [m, n_t] -> (region index [m] (-1: none), x [m, n_x] or None).  ``inclusive``: membership is
        ``E theta <= f + tol`` (MPC_LOCATE_INCLUSIVE) instead of the strict ``E theta - f < tol``."""
        th = _f64(theta).reshape(-1, self.n_t)
        m = len(th)
        region = numpy.empty(m, dtype=numpy.int64)
        x = numpy.empty((m, self.n_x)) if want_x else None
        ms = ctypes.c_float(0.0)
        rc = self._L.mpc_locator_query(self._h, m, th.ctypes.data_as(_dp), float(tol),
                                       (MPC_LOCATE_OVERLAPPING if overlapping else 0) | (MPC_LOCATE_INCLUSIVE if inclusive else 0)
                                       | (MPC_LOCATE_WALK if walk and self.has_adjacency else 0),
                                       region.ctypes.data_as(_lp), None if x is None else x.ctypes.data_as(_dp), ctypes.byref(ms))
        if rc != MPC_OK:
            raise MpcError(f'mpc_locator_query failed ({rc}): {self._L.mpc_last_global_error().decode()}')
        self.last_ms = float(ms.value)
        return region, x

    def close(self):
        if getattr(self, '_h', None):
            self._L.mpc_locator_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
