"""fp32 master parameters + bf16 shadow copies for the GEMM/conv operands.

The reference trains with apex O1 (``mmdet/apis/train.py:82-89``): fp32 master weights, half-precision
GEMM inputs, fp32 gradients.  Casting every weight on every use costs three tiny kernels per parameter
per step (cast, cast-backward, gradient accumulate); here all shadows live in ONE flat bf16 buffer that is
refreshed from the masters with a single multi-tensor copy after the optimizer step, and the shadows are
the autograd leaves (their bf16 gradients are gathered into the flat fp32 gradient buckets by ``ddp.py``).
"""
import contextlib
import ctypes
import os

import torch

from ._lib import half_dtype as _H

_SHADOW = {}          # id(master parameter) -> bf16 leaf tensor (a view of the flat shadow buffer)
_CONST = {}           # id(master parameter) -> bf16 constant copy (no gradient; biases of shadowed layers)
_SINK = {}            # id(master parameter) -> (fp32 gradient view in a flat bucket, notify())  -- set by ddp.py


def grad_sink(p):
    """Where a weight-gradient kernel may ACCUMULATE the fp32 gradient of master parameter ``p`` directly
    (the flat all-reduce bucket), plus the callback that tells the reducer the gradient has arrived.
    None when no reducer is active: the op then returns the gradient through autograd as usual."""
    return None if p is None else _SINK.get(id(p))


def register_sinks(table):
    _SINK.update(table)


def clear_sinks(ids=None):
    if ids is None:
        _SINK.clear()
    else:
        for i in ids:
            _SINK.pop(i, None)


# ---- second HIP stream for the weight-gradient kernels -------------------------------------------------------------
# Weight gradients are leaves of the backward dependency graph: nothing on the device consumes them before the
# all-reduce / optimizer.  Most backward kernels of this model are latency-bound (one block per CU, 20-40 us), so the
# weight-gradient launches go to a second stream where they fill the idle issue slots of the data-gradient chain
# (DESIGN section 5 "two streams").  Ordering: the side stream waits for an event recorded on the main stream after
# the producing kernel (fork); the main stream waits for the side stream before a bucket is reduced and at
# reducer.finish() (join).  Tensors the side stream reads are kept alive until the join (side_keep).
_SIDE_ON = os.environ.get("SWIN_WGRAD_STREAM", "1") != "0"      # 0: everything on one stream (A/B)
_LOW_PRIO = os.environ.get("SWIN_SIDE_PRIORITY", "low") == "low"      # "normal": a plain stream (A/B)
_SIDE = {}            # (device index, kind) -> torch.cuda.Stream; kind 'side': weight gradients and reductions (work nothing waits
                      # for), 'branch': data-dependent work of an independent sub-graph (it must not queue behind the former)
_SIDE_DIRTY = set()   # (device index, kind) with work since the last join


def side_enabled():
    return _SIDE_ON


def set_side_enabled(on):
    global _SIDE_ON
    side_join()
    _SIDE_ON = bool(on)


def _dev_index(device):
    return device.index if device.index is not None else torch.cuda.current_device()


def side_stream(device, kind='side'):
    """The weight-gradient stream (or, kind='branch', the sub-graph stream) of ``device``, created on first use; None when the
    feature is off."""
    if not _SIDE_ON or device.type != 'cuda':
        return None
    k = (_dev_index(device), kind)
    s = _SIDE.get(k)
    if s is None:
        s = None
        if kind == 'side' and _LOW_PRIO:
            # work nothing waits for: the lowest stream priority, so that it fills idle CUs instead of competing with the main stream
            try:
                from ._lib import call
                h = ctypes.c_void_p()
                with torch.cuda.device(k[0]):
                    call("swin_stream_create_low_priority", ctypes.byref(h))
                s = torch.cuda.ExternalStream(h.value, device=k[0])
            except Exception:
                s = None
        if s is None:
            s = torch.cuda.Stream(device=k[0])
        _SIDE[k] = s
    return s


def side_protect(device, *tensors, kind='side'):
    """Tell the caching allocator that the side stream reads these tensors (their memory is not handed out again until
    the side stream has passed the point of their release)."""
    s = side_stream(device, kind)
    if s is None:
        return
    for t in tensors:
        if t is not None and t.is_cuda:
            t.record_stream(s)


def side_mark(device, kind='side'):
    _SIDE_DIRTY.add((_dev_index(device), kind))


# Tensors an auxiliary stream reads are kept alive HERE until the next join instead of being handed to the allocator with
# Tensor.record_stream: that costs an event per freed block plus polling -- with ~10 tensors per Swin block it made the step
# host-bound (tools/host_time.py: 13.8 ms of host work per step against 11.7 ms on one stream).  After the join the main
# stream has waited for the auxiliary ones, so dropping the references (the memory returns to the main stream's pool) is safe.
_SIDE_KEEP = []


def side_keep(*tensors):
    _SIDE_KEEP.extend(t for t in tensors if t is not None)


_FORK_EVENTS = {}     # (device index, slot) -> torch.cuda.Event, reused round-robin (creating one per fork costs ~10 us of host time)
_FORK_NEXT = [0]


def _fork(cur, s, i):
    """``s`` waits for everything enqueued on ``cur`` so far."""
    k = (i, _FORK_NEXT[0] & 1023)
    _FORK_NEXT[0] += 1
    ev = _FORK_EVENTS.get(k)
    if ev is None:
        ev = _FORK_EVENTS[k] = torch.cuda.Event()
    ev.record(cur)
    s.wait_event(ev)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _raw_current(i):
    """handle (int) of device i's current stream"""
    if _raw_stream is not None:
        return _raw_stream(i)
    return torch.cuda.current_stream(i).cuda_stream


def fork_to_side(device, *tensors):
    """For a launch through the C ABI that nothing on the current stream waits for (a weight gradient): make the side stream
    wait for the current one, keep ``tensors`` (what the launch reads) alive until the next join, and return the side stream's
    handle to pass as the call's stream argument -- or None when the feature is off (use the current stream)."""
    s = side_stream(device)
    if s is None:
        return None
    i = _dev_index(device)
    cur, sp = _raw_current(i), s.cuda_stream
    if cur != sp:
        from ._lib import call
        call("swin_fork_stream", ctypes.c_void_p(cur), ctypes.c_void_p(sp))      # event ring in the library: no torch objects
    side_keep(*tensors)
    _SIDE_DIRTY.add((i, 'side'))
    return sp


@contextlib.contextmanager
def on_side(device, *tensors, kind='side'):
    """Run the enclosed launches on the weight-gradient stream (kind='branch': the sub-graph stream), after everything
    enqueued so far on the current stream; ``tensors`` = what they read.  No-op (current stream) when the feature is off.
    Reducer callbacks must be made OUTSIDE this context (they may enqueue collectives relative to the current stream)."""
    s = side_stream(device, kind)
    if s is None:
        yield None
        return
    cur = torch.cuda.current_stream(device)
    if cur != s:
        _fork(cur, s, _dev_index(device))
    side_keep(*tensors)
    side_mark(device, kind)
    with torch.cuda.stream(s):
        yield s


# Off by default: measured round 2 (bench.py, same box) 12.53 ms per step without, 12.93 ms with the small levels on their own
# stream (and 13.6 ms when they shared the weight-gradient stream, queueing behind its backlog in backward) -- the engine's
# cross-stream events and the extra record_stream / wait_stream host work cost more than the idle CUs were worth.
_SMALL_ON = os.environ.get("SWIN_SMALL_LEVELS_SIDE", "0") == "1"


def small_branch(x, limit=16384):
    """Context for the work of ONE small pyramid level (N*H*W <= limit pixels: P4-P6 at 800x1280) in training: its 3x3 convs
    and head GEMMs occupy 10-126 of the 256 CUs for 40 us each, so they run on the second stream next to the big levels'
    kernels on the main one -- a stream of its own: in backward these are data gradients the FPN waits for, which must not
    queue behind the weight-gradient stream's backlog (measured: sharing that stream cost 1.2 ms per step).  Autograd runs
    their backward on the same stream.  Yields the stream, or None when the level stays on the main stream.  The caller hands results that the main stream will read to ``side_outputs`` and joins
    (``side_join``) before reading them."""
    if (_SMALL_ON and _SIDE_ON and x.is_cuda and torch.is_grad_enabled() and x.dim() == 4
            and x.shape[0] * x.shape[2] * x.shape[3] <= limit):
        return on_side(x.device, x, kind='branch')
    return contextlib.nullcontext()


def side_outputs(*tensors):
    """Tensors produced on the second stream that the current (main) stream will read: their memory must not return to the
    second stream's pool while main-stream work still uses it."""
    for t in tensors:
        if t is not None and t.is_cuda:
            t.record_stream(torch.cuda.current_stream(t.device))


def fork_into(device, target, home=None):
    """``target`` (a torch stream) waits for everything enqueued so far on ``home`` (default: the current stream) and on every
    auxiliary stream with outstanding work; none of them waits for anything.  Marks ``target`` as having work to join."""
    from ._lib import call
    wgrad_flush()                       # recorded weight gradients write the buckets this fork is about to gather
    i = _dev_index(device)
    tp = target.cuda_stream
    hp = home.cuda_stream if home is not None else _raw_current(i)
    if hp != tp:
        call("swin_fork_stream", ctypes.c_void_p(hp), ctypes.c_void_p(tp))
    for k in _SIDE_DIRTY:
        sp = _SIDE[k].cuda_stream
        if k[0] == i and sp != tp and sp != hp:
            call("swin_fork_stream", ctypes.c_void_p(sp), ctypes.c_void_p(tp))
    for k, st in _SIDE.items():
        if st is target:
            _SIDE_DIRTY.add(k)


def host_gc_for_training(freeze=True):
    """Host-side setting for an eager training loop, called once after warm-up: the step creates a few thousand short-lived Python
    objects (tensors, ctypes arguments, autograd nodes), enough for a generation-0 collection every few launches and a full
    collection over the whole model every few steps.  Freeze what exists (the model, the plans: never garbage) and raise the
    thresholds; reference cycles are still collected, just less often.  Returns the previous thresholds."""
    import gc
    old = gc.get_threshold()
    gc.collect()
    if freeze and hasattr(gc, "freeze"):
        gc.freeze()
    gc.set_threshold(50000, 20, 100)
    return old


# ---- grouped Linear weight gradients (csrc/wgrad_dma.hip: swin_wgrad_record / swin_wgrad_flush) ----------------------------------
# Inside a training step with gradient sinks (ddp.BucketedGradReducer) the Linear layers do not launch their weight-gradient GEMM in
# their own backward: they RECORD it, and a stage's worth is launched as one grouped kernel -- hundreds of output tiles per launch, so
# no (or few) splits of the token axis and few float atomics, long loops that stream, and ~60 launches fewer per step.  Recorded
# operands are kept alive here until the flush; every join of the streams and every bucket launch flushes first.
_WG_ON = os.environ.get("SWIN_WGRAD_GROUP", "1") != "0"      # 0: every weight gradient in its own launch (A/B)
_WG_ENABLED = [0]     # > 0 while a reducer with gradient sinks is alive (someone flushes before the gradients are read)
_WG_PENDING = {}      # device index -> [problems, 128x128 tiles]
_WG_KEEP = []
_WG_STREAMS = {}      # device index -> raw handles of the streams the recorded operands were produced on
_WG_FLUSH_TILES = int(os.environ.get("SWIN_WGRAD_GROUP_TILES", "600"))
_WG_TAIL = int(os.environ.get("SWIN_WGRAD_TAIL_FLUSH", "0"))      # 1: launch the narrow stages' gradients block by block (A/B: slower);
                                                                    # 2: one extra launch at the stage 2 / stage 1 boundary


def wgrad_group_active():
    return _WG_ON and _WG_ENABLED[0] > 0


def wgrad_group_enable(on):
    """Reference-counted by the reducers: recording is only safe while someone will flush before the gradients are read."""
    if not on:
        wgrad_flush()
    _WG_ENABLED[0] = max(0, _WG_ENABLED[0] + (1 if on else -1))


def wgrad_record(dy2, x2, dwf, dbf, *keep):
    """Record dw += dy2^T x2 (db += column sums of dy2) for the next grouped launch; keeps the operands alive."""
    from ._lib import call
    N1, N2 = dy2.shape[1], x2.shape[1]
    call("swin_wgrad_record", ctypes.c_void_p(dy2.data_ptr()), ctypes.c_void_p(x2.data_ptr()), ctypes.c_void_p(dwf.data_ptr()),
         ctypes.c_void_p(dbf.data_ptr()) if dbf is not None else None, dy2.shape[0], N1, N2)
    wgrad_note(dy2.device, 1, ((N1 + 127) // 128) * ((N2 + 127) // 128), dy2, x2, dwf, dbf, *keep)


def wgrad_note(device, problems, tiles, *keep, tail=False):
    """``problems`` weight gradients (``tiles`` output tiles) were recorded on ``device`` (by wgrad_record or by swin_block_bwd);
    ``keep``: what they read and write.  Flushes once a launch's worth has accumulated.
    ``tail`` (with SWIN_WGRAD_TAIL_FLUSH=1 and the weight-gradient stream on): flush at once.  The blocks of the two narrow stages
    are the END of backward: nothing is left to hide their grouped launch behind and the optimizer waits for it (0.35 ms in the eager
    trace).  Launched block by block they do overlap the rest of backward -- but stage 1's backward and its weight gradients are both
    HBM streams, so the overlap only shares the bandwidth: measured 9.91 ms per step against 9.73 with the grouped tail; one extra launch
    at the stage 2 / stage 1 boundary (SWIN_WGRAD_TAIL_FLUSH=2, ``tail`` == 2): 9.99 against 9.89, twice.  Off."""
    i = _dev_index(device)
    ent = _WG_PENDING.setdefault(i, [0, 0])
    ent[0] += problems
    ent[1] += tiles
    _WG_KEEP.extend(t for t in keep if t is not None)
    # the operands come from the stream this backward node runs on: the main stream, or the sub-graph stream for the box head
    # (detector._roi_stage_train_packed) -- the grouped launch must wait for every one of them, not only for the stream that
    # happens to trigger the flush (found by test_graph_replay_equals_eager_steps)
    _WG_STREAMS.setdefault(i, set()).add(_raw_current(i))
    if ent[1] >= _WG_FLUSH_TILES or ent[0] >= 30 or (tail and _SIDE_ON and (_WG_TAIL == 1 or (_WG_TAIL == 2 and tail == 2))):
        wgrad_flush()


def wgrad_flush():
    """Launch everything recorded: on the weight-gradient stream behind the current one (fork), or on the current stream."""
    if not _WG_PENDING:
        return
    from ._lib import call
    for i, ent in list(_WG_PENDING.items()):
        if ent[0] == 0:
            continue
        dev = torch.device("cuda", i)
        with torch.cuda.device(i):
            sp = fork_to_side(dev)                       # None: feature off -> the current stream
            target = sp if sp is not None else _raw_current(i)
            for h in _WG_STREAMS.get(i, ()):             # producers on other streams than the one the fork came from
                if h != target and h != _raw_current(i):
                    call("swin_fork_stream", ctypes.c_void_p(h), ctypes.c_void_p(target))
            call("swin_wgrad_flush", ctypes.c_void_p(target))
        if sp is not None:
            side_keep(*_WG_KEEP)                         # the side stream reads them: alive until the next join
    _WG_PENDING.clear()
    _WG_KEEP.clear()
    _WG_STREAMS.clear()


def side_join():
    """The current stream of every device with outstanding side-stream work waits for it."""
    wgrad_flush()
    if _SIDE_DIRTY:
        from ._lib import call
        for k in _SIDE_DIRTY:
            cur, sp = _raw_current(k[0]), _SIDE[k].cuda_stream
            if cur != sp:
                call("swin_fork_stream", ctypes.c_void_p(sp), ctypes.c_void_p(cur))
        _SIDE_DIRTY.clear()
    _SIDE_KEEP.clear()


# ---- per-step caches and use counts ------------------------------------------------------------------------------
_DERIVED = {}         # (id(master), tag) -> tensor derived from the shadow (re-laid-out conv weights ...); cleared by refresh()
_STEPBUF = {}         # (id(master), tag) -> persistent scratch buffer
_STEPLIVE = set()     # buffers already zeroed in the current accumulation
_USES = {}            # id(master) -> forward uses whose backward has not run yet
_PENDING = {}         # id(master) -> finalizer to run at reducer.finish() if the count never reached zero


def derived(master, tag, fn):
    """fn() cached until the next ShadowParams.refresh() when ``master`` has a shadow (its value is then constant
    over the step); recomputed on every call otherwise."""
    if master is None or id(master) not in _SHADOW:
        return fn()
    k = (id(master), tag)
    v = _DERIVED.get(k)
    if v is None:
        v = _DERIVED[k] = fn()
    return v


def step_buffer(master, tag, shape, device):
    """A persistent fp32 scratch accumulator for ``master``; zero at its first request of an accumulation round (all known
    buffers are zeroed together by ONE multi-tensor launch in reset_step(); a buffer created later zeroes itself)."""
    k = (id(master), tag)
    b = _STEPBUF.get(k)
    if b is None or tuple(b.shape) != tuple(shape) or b.device != device:
        b = _STEPBUF[k] = torch.empty(shape, device=device, dtype=torch.float32)
        _STEPLIVE.discard(k)
    if k not in _STEPLIVE:
        b.zero_()
        _STEPLIVE.add(k)
    return b


def step_buffer_done(master, tag):
    _STEPLIVE.discard((id(master), tag))


def use_begin(master):
    _USES[id(master)] = _USES.get(id(master), 0) + 1


def use_end(master):
    """-> forward uses of ``master`` still waiting for their backward (0: this was the last one)."""
    n = max(_USES.get(id(master), 1) - 1, 0)
    _USES[id(master)] = n
    return n


def set_pending(master, fn):
    if fn is None:
        _PENDING.pop(id(master), None)
    else:
        _PENDING[id(master)] = fn


def flush_pending():
    """Run the finalizers of parameters whose last backward never came (a use outside the loss's graph)."""
    for k in list(_PENDING):
        _PENDING.pop(k)()


# ---- data gradients that already carry the ReLU backward of the layer below ------------------------------------------------
_GATED = {}           # data_ptr of a live data-gradient tensor -> (the tensor (kept alive: its address stays unique), gate data_ptr)


def gated_mark(dx, gate):
    """``dx`` was produced already zeroed where ``gate`` (the ReLU output that fed the producing layer) is not positive."""
    if len(_GATED) > 64:          # marks nobody took (a consumer that is not one of this package's convs): do not pile up tensors
        _GATED.clear()
    _GATED[dx.data_ptr()] = (dx, gate.data_ptr(), gate.numel())


def gated_take(dy, y):
    """True when ``dy`` is exactly such a tensor for ReLU output ``y`` (the caller may then skip threshold_backward).  A
    gradient that autograd summed from several consumers is a new tensor and is not found: the caller masks it as usual
    (masking is idempotent, so the already-masked contribution stays correct)."""
    ent = _GATED.pop(dy.data_ptr(), None)
    return ent is not None and ent[1] == y.data_ptr() and ent[2] == y.numel() and ent[0].numel() == dy.numel()


def reset_step():
    _USES.clear()
    _PENDING.clear()
    _STEPLIVE.clear()
    _GATED.clear()
    if _STEPBUF:                                  # one launch for all of them instead of one memset per layer
        bufs = [b for b in _STEPBUF.values() if b.is_cuda]
        if bufs:
            torch._foreach_zero_(bufs)
            _STEPLIVE.update(k for k, b in _STEPBUF.items() if b.is_cuda)


def weight(p, dtype):
    """The tensor a GEMM/conv should consume for parameter ``p`` in compute dtype ``dtype``."""
    if p is None or p.dtype == dtype:
        return p
    s = _SHADOW.get(id(p))
    if s is not None and s.dtype == dtype:
        return s
    return p.to(dtype)


def leaf(p):
    """The autograd leaf that stands for master parameter ``p``: its compute-dtype shadow when it has one, else ``p``."""
    return _SHADOW.get(id(p), p)


def shadow_of(p):
    """The bf16 copy an optimizer should refresh together with ``p`` (leaf shadow or constant), or None."""
    s = _SHADOW.get(id(p))
    return s if s is not None else _CONST.get(id(p))


def shadows_refreshed():
    """Called by an optimizer that has rewritten the shadows itself: drops the per-step derived tensors."""
    _DERIVED.clear()
    refresh_aux()


def const(p, dtype):
    """A compute-dtype copy of ``p`` that is NOT part of the autograd graph (None if there is none): the caller
    must deliver the gradient of ``p`` itself."""
    if p is None:
        return None
    c = _CONST.get(id(p))
    return c if c is not None and c.dtype == dtype else None


_LIVE = []            # weak references to the ShadowParams objects alive (refresh_all)


def refresh_all():
    """Re-copy every live ShadowParams' bf16 shadows / constants from the fp32 masters and drop the derived per-step tensors:
    for code that rewrites the masters behind the optimizer's back (checkpoint loading after the shadows exist)."""
    for r in list(_LIVE):
        sp = r()
        if sp is None:
            _LIVE.remove(r)
        else:
            sp.refresh()
    _DERIVED.clear()
    refresh_aux()


def is_khwc(p):
    """True for a 4-D tensor whose memory is (N, H, W, C)-contiguous ("channels last"), the layout the 3x3 conv kernels
    read weights in, and that is not also plainly contiguous."""
    return p.dim() == 4 and not p.is_contiguous() and p.is_contiguous(memory_format=torch.channels_last)


def dense_view(flat, off, p):
    """A view of flat[off : off + p.numel()] with p's shape AND p's memory layout (contiguous or channels-last), so that
    parameter, gradient bucket, optimizer state and bf16 shadow correspond element by element in memory."""
    seg = flat[off:off + p.numel()]
    if p.is_contiguous():
        return seg.view_as(p)
    if is_khwc(p):
        n, c, h, w = p.shape
        return seg.view(n, h, w, c).permute(0, 3, 1, 2)
    raise ValueError(f"parameter of shape {tuple(p.shape)} with strides {p.stride()} is neither contiguous nor channels-last")


@torch.no_grad()
def khwc_resident_(module):
    """Keep the weights of the 3x3 convolutions that run on csrc/conv_gemm.hip RESIDENT in the kernels' (Cout, ky, kx, Cin)
    memory layout: the parameter keeps its (Cout, Cin, 3, 3) shape (state_dict, checkpoints and the reference's init code see
    no difference) with channels-last strides.  Shadow, gradient bucket and optimizer state follow the parameter's layout
    (dense_view), so the forward needs no re-layout copy and the weight-gradient kernel accumulates straight into the
    all-reduce bucket.  Call before ShadowParams / the reducer / the optimizer are built."""
    import torch.nn as nn
    n = 0
    for m in module.modules():
        if (isinstance(m, nn.Conv2d) and m.kernel_size == (3, 3) and m.stride == (1, 1) and m.padding == (1, 1) and m.groups == 1
                and m.in_channels % 64 == 0 and m.out_channels % 64 == 0 and m.weight.is_contiguous()):
            m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
            n += 1
    return n


# ---- second compute-dtype layouts of shadowed weights, rebuilt for all of them by ONE launch after an optimizer step -----------
_AUX = {}             # (id(master), tag) -> (shadow, buffer)


def conv_dgrad_weight(master, shadow):
    """The (Cin, 3, 3, Cout) weight of the data-gradient convolution -- rot180, in/out swapped -- of a khwc-resident 3x3 conv
    weight with a bf16 shadow: a persistent buffer that refresh_aux() rewrites for every registered weight in one launch."""
    k = (id(master), 'dgrad')
    v = _DERIVED.get(k)
    if v is not None:
        return v
    ent = _AUX.get(k)
    if ent is None or ent[0] is not shadow:
        co, ci = shadow.shape[0], shadow.shape[1]
        _AUX[k] = (shadow, torch.empty(ci, 3, 3, co, device=shadow.device, dtype=shadow.dtype))
    refresh_aux()
    return _DERIVED[k]


def linear_t_weight(master, shadow):
    """The transposed (in, out) copy of a Linear weight's bf16 shadow -- the K-contiguous operand of its data-gradient GEMM on the
    hand-written kernel (swin_linear_dgelu_hip_bf16): a persistent buffer that refresh_aux() rewrites for every registered weight
    in one launch after each optimizer step.  Without a shadow (no ShadowParams): a plain transposed copy."""
    if master is None or id(master) not in _SHADOW:
        return shadow.detach().t().contiguous()
    k = (id(master), 't')
    v = _DERIVED.get(k)
    if v is not None:
        return v
    ent = _AUX.get(k)
    if ent is None or ent[0] is not shadow:
        _AUX[k] = (shadow, torch.empty(shadow.shape[1], shadow.shape[0], device=shadow.device, dtype=shadow.dtype))
    refresh_aux()
    return _DERIVED[k]


def rel_bias_expanded(table, gate_master):
    """The (nH, 64, 64) expansion of a relative-position-bias table (swin_transformer.py:105-110 with the 49 -> 64 padding mask
    folded in) as a persistent buffer that refresh_aux() rebuilds for every registered table in one launch after each optimizer
    step.  Only while ``gate_master`` (a weight of the same block) has a shadow -- the table then changes exactly when the shadows
    are refreshed; returns None otherwise (the caller expands the table itself)."""
    if gate_master is None or id(gate_master) not in _SHADOW or not table.is_cuda or table.dtype != torch.float32:
        return None
    k = (id(table), 'relbias')
    v = _DERIVED.get(k)
    if v is not None:
        return v
    ent = _AUX.get(k)
    if ent is None or ent[0] is not table:
        _AUX[k] = (table, torch.empty(table.shape[1], 64, 64, device=table.device, dtype=torch.float32))
    refresh_aux()
    return _DERIVED[k]


def refresh_aux():
    """Rebuild every registered auxiliary layout from the current shadows (one multi-tensor launch per kind) and publish them."""
    if not _AUX:
        return
    from .ops.functional import conv_dgrad_layout_multi, linear_t_layout_multi, rel_bias_expand_multi
    conv = [(k, e) for k, e in _AUX.items() if k[1] == 'dgrad']
    lin = [(k, e) for k, e in _AUX.items() if k[1] == 't']
    rel = [(k, e) for k, e in _AUX.items() if k[1] == 'relbias']
    if conv:
        conv_dgrad_layout_multi([e[0] for _, e in conv], [e[1] for _, e in conv])
    if lin:
        linear_t_layout_multi([e[0] for _, e in lin], [e[1] for _, e in lin])
    if rel:
        rel_bias_expand_multi([e[0].detach() for _, e in rel], [e[1] for _, e in rel])
    for k, e in conv + lin + rel:
        _DERIVED[k] = e[1]


class ShadowParams:
    def __init__(self, module, dtype=None, min_numel=1024):
        """Shadows are created for the ``weight`` (>= min_numel elements) and ``bias`` of Linear / Conv2d /
        ConvTranspose2d layers (the GEMM / conv operands).  Everything else (LayerNorm affine, relative-position
        bias tables, position embeddings) is consumed in fp32 by the kernels and keeps its direct gradient."""
        import torch.nn as nn
        dtype = _H() if dtype is None else dtype
        self.dtype = dtype
        seen, self.masters, self.const_masters = set(), [], []
        for m in module.modules():
            if isinstance(m, (nn.Linear, nn.Conv2d, nn.ConvTranspose2d)):
                p = m.weight
                if p.requires_grad and p.dtype == torch.float32 and p.numel() >= min_numel and id(p) not in seen:
                    seen.add(id(p))
                    self.masters.append(p)
                    # the bias of a shadowed layer gets a bf16 CONSTANT copy for the GEMM epilogue (addmm): not an
                    # autograd leaf -- ops.linear hands the bias gradient to the fp32 master itself; the fused
                    # kernels that want the bias in fp32 (bias_gelu, attention pad rows, conv3x3) read the master
                    b = m.bias
                    if b is not None and b.dtype == torch.float32 and id(b) not in seen:
                        seen.add(id(b))
                        self.const_masters.append(b)
        pad = lambda k: (k + 7) // 8 * 8                       # 16-byte aligned slots
        n = sum(pad(p.numel()) for p in self.masters + self.const_masters)
        dev = self.masters[0].device if self.masters else torch.device("cpu")
        self.flat = torch.empty(n, device=dev, dtype=dtype)
        self.shadows, self.consts = [], []
        off = 0
        for p in self.masters:
            v = dense_view(self.flat, off, p)
            v.requires_grad_(True)
            self.shadows.append(v)
            _SHADOW[id(p)] = v
            off += pad(p.numel())
        for p in self.const_masters:
            v = dense_view(self.flat, off, p)
            self.consts.append(v)
            _CONST[id(p)] = v
            off += pad(p.numel())
        import weakref
        _LIVE.append(weakref.ref(self))
        self.refresh()

    @torch.no_grad()
    def refresh(self):
        """shadow <- master (after every optimizer step): one multi-tensor copy."""
        if self.masters:
            torch._foreach_copy_(self.shadows + self.consts, self.masters + self.const_masters)
        _DERIVED.clear()
        refresh_aux()

    def leaf_of(self, p):
        return _SHADOW.get(id(p), p)

    def release(self):
        for p in self.masters:
            _SHADOW.pop(id(p), None)
        for p in self.const_masters:
            _CONST.pop(id(p), None)
        for p in self.masters:
            _AUX.pop((id(p), 'dgrad'), None)
            _AUX.pop((id(p), 't'), None)
        for k in [k for k in _AUX if k[1] == 'relbias']:      # gated on a shadowed weight of their block: none survives the shadows
            _AUX.pop(k, None)
        _DERIVED.clear()


class LossScaler:
    """Dynamic loss scaling for the fp16 build, entirely on the device (apex O1's dynamic LossScaler, which the reference
    trains with: mmdet/apis/train.py:82-89): the scale, the skip flag and the clean-step counter live in the optimizer's
    device-resident state, so a step never waits for the host and the whole step can sit inside a captured hipGraph.

        scaler = LossScaler(optim, reducer)
        loss = scaler.scale(loss); loss.backward(); reducer.finish()
        scaler.check(optim); optim.step(); scaler.update(optim)

    check(): non-finite gradient anywhere -> the optimizer launch leaves parameters and moments untouched; every gradient is
    multiplied by 1 / scale inside the optimizer kernel.  update(): halve on a skipped step, double after ``growth_interval``
    clean ones."""

    def __init__(self, optim, reducer, init_scale=2.0 ** 16, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000,
                 min_scale=1.0, max_scale=2.0 ** 24):
        self.optim, self.reducer = optim, reducer
        self.growth, self.backoff, self.interval = float(growth_factor), float(backoff_factor), int(growth_interval)
        self.min_scale, self.max_scale = float(min_scale), float(max_scale)
        dev = reducer.buckets[0]['flat'].device
        self.state = optim.state_tensor(dev)
        self.state[20] = float(init_scale)
        self.state[21] = 0.0
        self._scale_view = self.state[20]                 # 0-dim view: the loss is multiplied by it on the device

    def scale(self, loss):
        return loss * self._scale_view

    def _sp(self):
        return ctypes.c_void_p(self.state.data_ptr())

    def check(self, optim=None):
        from ._lib import call
        st = ctypes.c_void_p(_raw_current(_dev_index(self.state.device)))
        call("swin_loss_scale_begin", self._sp(), st)
        for b in self.reducer.buckets:
            f = b['flat']
            call("swin_grad_check_finite", ctypes.c_void_p(f.data_ptr()), f.numel(), self._sp(), st)

    def update(self, optim=None):
        from ._lib import call
        st = ctypes.c_void_p(_raw_current(_dev_index(self.state.device)))
        call("swin_loss_scale_update", self._sp(), self.growth, self.backoff, self.interval, self.min_scale, self.max_scale, st)

    def get_scale(self):
        """host copy of the current scale (synchronises: logging / tests only)"""
        return float(self.state[20])

    def skipped_last_step(self):
        return float(self.state[19]) != 0.0
