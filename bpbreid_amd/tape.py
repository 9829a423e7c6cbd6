"""Launch tape: record the C-ABI calls of one train step once, replay them with one host call per segment.

The backbone of a step is a static launch plan (graph.Net, csrc/plan.cpp).  The rest -- part-attention head, dense stack, GiLt and
pixel losses, their backward passes, Adam -- is ~150 calls of libbpbreid_hip.so whose arguments do not change between steps once
every buffer they touch lives as long as the plan.  `Tape` records those calls while they execute (native.call appends every call
that carries the launch stream) and `run()` replays them through `bpb_tape_run` (csrc/tape.cpp): no Python between the launches, no
allocator, no ctypes marshalling; Python callables recorded with `python()` (the hand-over of a gradient bucket to RCCL) run between
the segments, in order.  The reference drives the same stretch from Python (part_based_engine.py:77-130, GiLt_loss.py:45-119).

Unlike a hipGraph the replayed launches are ordinary eager launches: the two-stream schedule of bpb_plan_run2 and RCCL's streams
work unchanged; a replay can itself be captured into a hipGraph like any other sequence of launches.
"""
import contextlib
import ctypes as C

from . import native as nv


class Tape:
    def __init__(self):
        self.items = []            # ('ops', TapeOp array, count) | ('py', callable)
        self.keep = []             # host objects whose addresses the tape holds (descriptor arrays, pointer tables, tensors)
        self._cur = []
        self._sigs = {}
        self.launch_calls = 0

    # ------------------------------------------------------------------ recording
    def _signature(self, fn):
        sig = self._sigs.get(fn)
        if sig is None:
            buf = C.create_string_buffer(nv.TAPE_MAX_ARGS + 1)
            n = nv.lib().bpb_tape_signature(fn, buf)
            if n < 0:
                raise nv.NativeError('bpb_tape_signature(%d) failed' % fn)
            sig = self._sigs[fn] = buf.value.decode()
            assert len(sig) == n
        return sig

    def record(self, name, args):
        """Called by native.call for every call made while this tape records.  Calls without the launch stream (workspace-size
        queries, host-side routines) are not part of the step's launch sequence and are skipped."""
        if not any(isinstance(a, nv.StreamArg) for a in args):
            return
        fn = nv.lib().bpb_tape_function(name.encode())
        if fn < 0:
            raise nv.NativeError('%s cannot be recorded on a launch tape (csrc/tape.cpp: BPB_TAPE_FUNCTIONS)' % name)
        sig = self._signature(fn)
        if len(sig) != len(args):
            raise nv.NativeError('%s takes %d arguments, %d recorded' % (name, len(sig), len(args)))
        op = nv.TapeOp()
        op.fn, op.nargs, op.stream_mask = fn, len(args), 0
        for q, (kind, a) in enumerate(zip(sig, args)):
            w = op.a[q]
            if kind in 'ps':
                if isinstance(a, nv.StreamArg):
                    if kind != 's':
                        raise nv.NativeError('%s: argument %d is not a stream' % (name, q))
                    op.stream_mask |= 1 << q
                    w.p = a.value
                elif a is None:
                    w.p = None
                elif isinstance(a, int):
                    w.p = a
                elif isinstance(a, C.c_void_p):
                    w.p = a.value
                elif isinstance(a, (C.Array, C.Structure)):
                    self.keep.append(a)
                    w.p = C.addressof(a)
                elif hasattr(a, '_obj'):                       # ctypes.byref(x)
                    self.keep.append(a._obj)
                    w.p = C.addressof(a._obj)
                else:
                    raise nv.NativeError('%s: argument %d (%s) cannot be recorded' % (name, q, type(a).__name__))
            elif kind == 'i':
                w.i = int(a)
            elif kind == 'l':
                w.l = int(a)
            elif kind == 'f':
                w.f = float(a)
            elif kind == 'd':
                w.d = float(a)
            else:
                raise nv.NativeError('%s: unknown parameter kind %r' % (name, kind))
        self._cur.append(op)
        self.launch_calls += 1

    def _flush(self):
        if self._cur:
            arr = (nv.TapeOp * len(self._cur))(*self._cur)
            self.items.append(('ops', arr, len(self._cur)))
            self._cur = []

    def python(self, fn):
        """A host-side action between two segments of the tape (e.g. handing a gradient bucket to RCCL): called with no
        arguments at the same position of every replay."""
        self._flush()
        self.items.append(('py', fn, 0))

    def freeze(self):
        self._flush()
        return self

    # ------------------------------------------------------------------ replay
    def run(self):
        L = nv.lib()
        s = nv.stream()
        for kind, payload, n in self.items:
            if kind == 'ops':
                nv.check(L.bpb_tape_run(payload, n, s))
            else:
                payload()


@contextlib.contextmanager
def recording(tape):
    """Every stream-carrying native.call inside the block is executed and appended to `tape`."""
    if nv._recording is not None:
        raise nv.NativeError('a launch tape is already recording')
    nv._recording = tape
    try:
        yield tape
    finally:
        nv._recording = None
        tape.freeze()


@contextlib.contextmanager
def paused():
    """Launches inside the block are executed but NOT recorded (constants written once when the tape is built)."""
    t, nv._recording = nv._recording, None
    try:
        yield
    finally:
        nv._recording = t
