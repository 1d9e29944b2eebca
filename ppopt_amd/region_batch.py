"""Structure-of-arrays storage for the regions of one BFS level.

The device returns the regions of a level as three dense arrays (include/mpcombi.h, mpc_level_regions_compact).
``RegionBatch`` keeps them as they are; ``BatchCriticalRegion`` is a ``CriticalRegion`` whose fields are numpy
views / lists cut out of those arrays the first time they are read (and cached, and assignable, like ordinary
attributes).  A Solution with 10^4 regions is therefore 10^4 small Python objects and three arrays, not 6*10^4 arrays
built eagerly inside the solve.
"""
import contextlib
import gc
import itertools
import threading
from typing import List

import numpy

from .critical_region import CriticalRegion

try:      # host-side helper of Solution.materialize (csrc_host/fastmat.c); absent in an unbuilt tree: the Python loop does the same
    import os as _os
    if _os.environ.get('MPC_NO_FASTMAT', '0') == '1':
        raise ImportError('switched off')
    from . import _fastmat
except ImportError:
    _fastmat = None


_gc_lock = threading.Lock()
_gc_pauses = 0
_gc_was_enabled = False


@contextlib.contextmanager
def gc_paused():
    """The cycle collector is held while ``solve_many`` creates its region objects (re-entrant, any thread).  Such a call allocates 10^5
    small container objects that are not garbage and form no cycles; every 700 of them start a collection of the young generation,
    every tenth of those one of the next, and the full collections walk everything alive.  Reference counting frees objects as
    before; the collector runs again as soon as the outermost call returns (``_promote_young`` below: without a walk over the new
    objects).  ``MPC_KEEP_GC=1`` leaves the collector alone.  Round 6: ``solve()`` holds it too."""
    global _gc_pauses, _gc_was_enabled
    import os
    if os.environ.get('MPC_KEEP_GC', '0') == '1':
        yield
        return
    with _gc_lock:
        if _gc_pauses == 0:
            _gc_was_enabled = gc.isenabled()
            gc.disable()
        _gc_pauses += 1
    try:
        yield
    finally:
        with _gc_lock:
            _gc_pauses -= 1
            if _gc_pauses == 0 and _gc_was_enabled:
                _promote_young()
                gc.enable()


def _promote_young():
    """Called with the collector still held, at the end of the outermost pause.  The pause has left thousands of new container objects in
    the young generation; the first allocation after ``gc.enable()`` would start a collection that walks every one of them (10^4 regions:
    0.1-0.3 ms, 10^5: 5 ms -- as much as the collections the pause avoided), and the survivors are walked again when the middle generation
    is collected.  ``gc.freeze(); gc.unfreeze()`` splices all generations into the permanent one and that back into the OLDEST (two list
    operations, no object is visited): the new objects -- region views that hold two references and form no cycles -- are from now on
    looked at by full collections only, like anything that has lived for a while.  So is whatever else was young at that moment; objects
    the caller has frozen deliberately stay frozen (the step is skipped then).  ``MPC_GC_PROMOTE=0`` switches it off."""
    import os
    if os.environ.get('MPC_GC_PROMOTE', '1') == '0' or gc.get_freeze_count() != 0 or gc.get_count()[0] < 2000:
        return
    gc.freeze()
    gc.unfreeze()


class RegionBatch:
    def __init__(self, head_d: numpy.ndarray, head_i: numpy.ndarray, erows: numpy.ndarray, n_x: int, n_t: int, n_c: int,
                 n_tc: int, k: int, slots=None):
        self.hd, self.hi, self.er = head_d, head_i, erows
        # rows of head_d / head_i that are regions (mpc_level_regions_slots returns one slot per optimal candidate)
        self.slots = numpy.arange(len(head_d)) if slots is None else numpy.asarray(slots)
        self.n_x, self.n_t, self.n_c, self.n_tc, self.k = n_x, n_t, n_c, n_tc, k
        self.oA, self.ob = 0, n_x * n_t
        self.oC, self.od = self.ob + n_x, self.ob + n_x + k * n_t
        self.iact, self.iom = 8, 8 + k
        self.ila = self.iom + n_tc
        self.iri = self.ila + k
        self.irc = self.iri + (n_c - k)
        # set by the mixed-integer enumeration for all regions of the batch at once (CriticalRegion.y_fixation / y_indices / x_indices)
        self.y_fixation = self.y_indices = self.x_indices = None

    def __len__(self):
        return len(self.slots)

    def regions(self) -> List['BatchCriticalRegion']:
        return self.regions_of(self.slots.tolist())

    def regions_of(self, slots) -> List['BatchCriticalRegion']:
        """Region objects of the given slots (a streamed level hands its slots over chunk by chunk).  (Round 6: created by the C helper
        when it is built -- no Python-level ``__init__`` per region.)"""
        if _fastmat is not None and type(slots) is list:
            return _fastmat.make(BatchCriticalRegion, self, slots)
        return list(map(BatchCriticalRegion, itertools.repeat(self), slots))


class _Lazy:
    """NON-DATA descriptor (no ``__set__``): computes the field from the batch on first access and leaves it in the instance
    dictionary, which every later read -- and every assignment -- then uses directly, at the speed of a plain attribute (an entry of
    the instance dictionary takes precedence over a non-data descriptor; round 5: the data-descriptor form ran a Python-level
    ``__get__`` on every read, 15 ms to touch the fields of config 4's 9,432 regions once more)."""

    def __init__(self, fn):
        self.fn = fn
        self.key = fn.__name__

    def __get__(self, obj, cls):
        if obj is None:
            return self
        d = obj.__dict__
        if self.key not in d:
            d[self.key] = self.fn(obj)
        return d[self.key]


_FIELD_NAMES = ('A', 'b', 'C', 'd', 'E', 'f', 'active_set', 'omega_set', 'lambda_set', 'regular_set')
_FIELDS = frozenset(_FIELD_NAMES)


class BatchCriticalRegion(CriticalRegion):
    """A CriticalRegion backed by row ``j`` of a RegionBatch (same fields, same index conventions)."""
    # the two references live in slots: the instance dictionary (inherited from the dataclass) is only created when a field is
    # first read -- a solve creates 10^4 of these per step and the next step frees them
    __slots__ = ('_batch', '_j')

    def __init__(self, batch: RegionBatch, j: int):  # no dataclass __init__: fields come from the batch
        self._batch = batch
        self._j = j

    def _hdr(self):
        return self._batch.hi[self._j]

    @_Lazy
    def A(self):
        B = self._batch
        return B.hd[self._j, B.oA:B.ob].reshape(B.n_x, B.n_t)

    @_Lazy
    def b(self):
        B = self._batch
        return B.hd[self._j, B.ob:B.oC].reshape(B.n_x, 1)

    @_Lazy
    def C(self):
        B = self._batch
        return B.hd[self._j, B.oC:B.od].reshape(B.k, B.n_t)

    @_Lazy
    def d(self):
        B = self._batch
        return B.hd[self._j, B.od:B.od + B.k].reshape(B.k, 1)

    @_Lazy
    def E(self):
        B, h = self._batch, self._hdr()
        return B.er[h[6]:h[6] + h[2], 1:]

    @_Lazy
    def f(self):
        B, h = self._batch, self._hdr()
        return B.er[h[6]:h[6] + h[2], :1]

    @_Lazy
    def active_set(self):
        B = self._batch
        return B.hi[self._j, B.iact:B.iact + B.k].tolist()

    @_Lazy
    def omega_set(self):
        B, h = self._batch, self._hdr()
        return B.hi[self._j, B.iom:B.iom + h[3]].tolist()

    @_Lazy
    def lambda_set(self):
        B, h = self._batch, self._hdr()
        return B.hi[self._j, B.ila:B.ila + h[4]].tolist()

    @_Lazy
    def regular_set(self):
        B, h = self._batch, self._hdr()
        return [B.hi[self._j, B.iri:B.iri + h[5]].tolist(), B.hi[self._j, B.irc:B.irc + h[5]].tolist()]

    @_Lazy
    def y_fixation(self):
        return self._batch.y_fixation

    @_Lazy
    def y_indices(self):
        return self._batch.y_indices

    @_Lazy
    def x_indices(self):
        return self._batch.x_indices

    def materialize(self) -> 'BatchCriticalRegion':
        """Touches every field (so that nothing refers to the batch lazily any more)."""
        if not self.__dict__.keys() >= _FIELDS:      # (after Solution.materialize every field is there already: one set comparison)
            for name in _FIELD_NAMES:
                getattr(self, name)
        return self


def materialize_regions(regions) -> None:
    """Touches every field of many regions at once: the regions of one RegionBatch are cut out of its three arrays with a handful
    of array-wide operations (one reshape per matrix field, one ``tolist`` per index field) instead of ten small slices per
    region.  Regions that are ordinary ``CriticalRegion`` objects are left as they are."""
    with gc_paused():      # 10^5 small lists and views that are not garbage: the cycle collector would walk them again and again
        _materialize_groups(_group_by_batch(regions))


def _group_by_batch(regions):
    """{id(batch): (batch, [regions])}.  A solution's list is runs of regions of the same batch (a level, or a chunk of one): the runs are
    found with one array comparison instead of a dictionary operation per region."""
    import operator
    regions = regions if isinstance(regions, list) else list(regions)
    groups = {}
    try:
        batches = list(map(operator.attrgetter('_batch'), regions))
    except AttributeError:      # ordinary CriticalRegion objects among them: the general way
        for r in regions:
            if isinstance(r, BatchCriticalRegion):
                groups.setdefault(id(r._batch), (r._batch, []))[1].append(r)
        return groups
    n = len(batches)
    if n == 0:
        return groups
    ids = numpy.fromiter(map(id, batches), dtype=numpy.int64, count=n)
    cuts = [0] + (numpy.flatnonzero(ids[1:] != ids[:-1]) + 1).tolist() + [n]
    for a, b in zip(cuts[:-1], cuts[1:]):
        g = groups.get(int(ids[a]))
        if g is None:
            groups[int(ids[a])] = (batches[a], regions[a:b])
        else:
            g[1].extend(regions[a:b])
    return groups


def _materialize_groups(groups) -> None:
    import operator
    for B, regs in groups.values():
        n = len(regs)
        if n == 0:
            continue
        js = numpy.fromiter(map(operator.attrgetter('_j'), regs), dtype=numpy.int64, count=n)
        if _fastmat is not None:
            # (round 6) the loop below in C (csrc_host/fastmat.c, built by __graft_entry__.build): the same views, lists and dictionary
            # per region in ~0.9 us instead of ~1.9; this Python form stays for an unbuilt tree and for layouts the C loop declines
            try:
                _fastmat.fill(regs if isinstance(regs, list) else list(regs), B.hd, B.hi, B.er, js,
                              (B.n_x, B.n_t, B.k, B.oA, B.ob, B.oC, B.od, B.iact, B.iom, B.ila, B.iri, B.irc))
                continue
            except (ValueError, TypeError):
                pass
        j0 = int(js[0])
        # (round 6) the usual case -- a level's regions in slot order, one contiguous run -- takes the slot rows as a VIEW, not as a gathered copy
        if n == 1 or (int(js[-1]) - j0 == n - 1 and bool(numpy.all(js[1:] - js[:-1] == 1))):
            hd, hi = B.hd[j0:j0 + n], B.hi[j0:j0 + n]
        else:
            hd, hi = B.hd[js], B.hi[js]
        # one list of views per matrix field (cut in C by iterating the stacked array), one tolist per index field
        A = list(hd[:, B.oA:B.ob].reshape(-1, B.n_x, B.n_t))
        b = list(hd[:, B.ob:B.oC].reshape(-1, B.n_x, 1))
        C = list(hd[:, B.oC:B.od].reshape(-1, B.k, B.n_t))
        d = list(hd[:, B.od:B.od + B.k].reshape(-1, B.k, 1))
        lo = hi[:, 6].tolist()
        up = (hi[:, 6] + hi[:, 2]).tolist()
        n_om, n_la = hi[:, 3].tolist(), hi[:, 4].tolist()
        act = hi[:, B.iact:B.iact + B.k].tolist()
        # (only as many columns as the longest list of the batch: the padded widths are n_tc, k and n_c - k)
        w_om, w_la, w_re = (int(hi[:, c].max()) for c in (3, 4, 5))
        om = hi[:, B.iom:B.iom + w_om].tolist()
        la = hi[:, B.ila:B.ila + w_la].tolist()
        # the two lists of the regular set are long rows with short contents (n_c - k columns, a region's own rows in use): only the entries
        # in use become Python integers -- one flat list per field, cut by the running offsets
        used = numpy.arange(w_re, dtype=numpy.int32)[None, :] < hi[:, 5:6]
        ri_flat = hi[:, B.iri:B.iri + w_re][used].tolist()
        rc_flat = hi[:, B.irc:B.irc + w_re][used].tolist()
        r1 = numpy.cumsum(hi[:, 5], dtype=numpy.int64).tolist()
        r0 = [0] + r1[:-1]
        erE, erf = B.er[:, 1:], B.er[:, :1]
        # (one zip over the columns: no index operations inside the loop; a region nobody has read yet gets the dictionary itself)
        for r, Ai, bi, Ci, di, o0, o1, acti, omi, lai, q0, q1, n1, n2 in zip(regs, A, b, C, d, lo, up, act, om, la, r0, r1, n_om, n_la):
            fields = {'A': Ai, 'b': bi, 'C': Ci, 'd': di, 'E': erE[o0:o1], 'f': erf[o0:o1], 'active_set': acti,
                      'omega_set': omi[:n1], 'lambda_set': lai[:n2], 'regular_set': [ri_flat[q0:q1], rc_flat[q0:q1]]}
            dd = r.__dict__
            if dd:      # fields that were read (or assigned) before keep their values
                fields.update(dd)
                dd.update(fields)
            else:
                r.__dict__ = fields
