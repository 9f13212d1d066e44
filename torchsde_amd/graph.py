"""HIP-graph capture of a whole fixed-step solve (``options={"hip_graph": "auto" | True | False}``, default "auto").

``"auto"`` (the drop-in default): the second solve with the same structure on the same SDE object is captured and every
later one replays the graph -- provided the capture is legal (see the constraints below) AND safe to do silently: the
first solve runs eagerly and screened (`run_screened`: code that synchronises with the host, or uses operators other than
the elementwise / matmul / reduction family, is never captured -- a failed capture cannot be recovered from on this
stack), the Python-side state of the SDE object (plain attributes, tensor identities and strides, `training` flags,
class attributes, and what its functions read from closures and module globals: `python_state`) is part of the cache key,
the captured graph's first replay must reproduce the eager solve it was recorded beside bit for bit, its first two
replays in real use and then every 8th, 64th, 512th ... run beside the eager path and are compared with it,
very large states stay eager (launch overhead does not matter there and a graph pins a second memory pool), and any
failure along the way falls back to the eager path without a word. ``True``: capture on first use, warn when
impossible (the caller vouches for capture-safe code). ``False``: never.

A solve is thousands of short kernels (the user's ``f``/``g`` torch ops plus one fused step kernel per stage).
For small and medium batches the GPU finishes each of them faster than Python can issue the next one; the
reference is in the same regime and additionally syncs three times per step. Here the launch-only part of a
solve (``BaseSDESolver._run``) is captured ONCE into a HIP graph -- through torch's capture stream, which also
records the ``libtorchsde_amd.so`` launches because they are issued on torch's current stream -- and replayed
for every later solve with the same SDE object, shapes, method, time grid and Brownian structure.

What changes between solves is the initial state (copied into the graph's static input buffer) and the Brownian
seed: the kernels read the entropy from one device word (``tsde_noise_t.entropy_dev``) that is rewritten before
each replay, so a new ``BrownianInterval`` (new entropy) reuses the captured graph.

Three things are recorded this way:

* ``_CapturedSolve``          a forward solve without autograd (``sdeint`` under ``no_grad``, the forward pass of
                              ``sdeint_adjoint``), including solvers that carry state between steps;
* ``_CapturedBackward``       the whole backward sweep of ``sdeint_adjoint`` (``adjoint_options={"hip_graph": True}``);
* ``_CapturedTrainingSolve``  a forward solve recorded WITH its autograd graph plus the back-propagation through it
                              (``sdeint`` with gradients on), as two graphs sharing a memory pool.

Constraints (checked, with a loud fallback to the eager path otherwise): a native ``BrownianInterval`` without
pinned ``W``/``H``; every trainable tensor is ``y0`` or a parameter of the SDE module; the user's ``f``/``g`` must be
capture-safe torch code (static shapes, no host sync, no Python-side state that changes between solves). Graphs
are cached on the user's SDE object, keyed by the structure of the solve (at most ``_MAX_GRAPHS_PER_SDE``).
"""
import contextlib
import gc
import warnings

import torch
from torch.utils._python_dispatch import TorchDispatchMode

from .brownian import BrownianInterval

_CACHE_ATTR = "_tsde_hip_graphs"


_MAX_GRAPHS_PER_SDE = 16


@contextlib.contextmanager
def _no_gc():
    """No cyclic garbage collection while a capture is open: a collected object whose destructor touches the HIP
    runtime (another CUDAGraph, an event, a cached block) aborts the process when it runs mid-capture."""
    gc.collect()
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was_enabled:
            gc.enable()


class _GraphCache(dict):
    """The captured graphs of one SDE object, kept on that object. A copy of the module (`copy.deepcopy` for an EMA or
    a checkpoint, pickling, `torch.save`) must not try to copy HIP graphs: the copy starts with an empty cache and
    captures its own graphs (its parameters live in other storage anyway)."""

    def __deepcopy__(self, memo):
        return _GraphCache()

    def __reduce__(self):
        return (_GraphCache, ())


def _cache_of(base):
    cache = getattr(base, _CACHE_ATTR, None)
    if cache is None:
        cache = _GraphCache()
        setattr(base, _CACHE_ATTR, cache)
    return cache


def describe_cache(sde):
    """What the HIP-graph cache on `sde` holds, one line per structure: the answer to "why is my solve (not) replayed?"."""
    base = sde
    while hasattr(base, "_base_sde"):
        base = base._base_sde
    lines = []
    for sig, entry in getattr(base, _CACHE_ATTR, {}).items():
        if isinstance(entry, _StatesSeen):
            continue
        kind = next((x for x in sig if isinstance(x, str) and x not in ("auto", "auto-churn")), "?")
        mode = "auto" if sig and sig[0] in ("auto", "auto-churn") else "explicit"
        if isinstance(entry, _Refused):
            what = f"stays eager: {entry.reason}"
        elif isinstance(entry, _Seen):
            what = ("seen once, next solve records it (drift and diffusion independent: "
                    f"{'yes' if entry.independent else 'no'})")
        else:
            tuning = getattr(entry, "tuning", None)
            what = type(entry).__name__.lstrip("_") + (f", {tuning}" if tuning else "")
            nodes = [getattr(getattr(entry, name, None), "memset_nodes", None) for name in ("graph", "fwd_graph", "bwd_graph")]
            found, rewritten = (sum(n[k] for n in nodes if n) for k in (0, 1))
            if found:
                what += f", {rewritten} of {found} memset nodes rewritten as kernels"
        lines.append(f"[{mode}] {kind}: {what}")
    return lines


_MAX_PINNED_BYTES = 8 << 30      # of result buffers of recorded graphs kept on ONE SDE object


def _pinned_bytes(entry):
    """Rough size of what a cache entry keeps allocated (its result buffers; the graph's pool is a small multiple)."""
    outs = getattr(entry, "outputs", None)
    if outs is None:
        return 0
    try:
        return sum(o.numel() * o.element_size() for o in outs())
    except Exception:
        return 0


def _remember(cache, sig, captured):
    """Each captured graph pins its memory pool; a caller that keeps changing the structure (e.g. random `ts`) must
    not grow the cache without bound -- neither in entries nor in bytes: the oldest entries go first."""
    while sig not in cache and len(cache) >= _MAX_GRAPHS_PER_SDE:
        cache.pop(next(iter(cache)))
    cache[sig] = captured
    total = sum(_pinned_bytes(v) for v in cache.values())
    for key in list(cache):
        if total <= _MAX_PINNED_BYTES or key == sig:
            break
        total -= _pinned_bytes(cache.pop(key))


# ---- "auto" mode ------------------------------------------------------------------------------------------------------
_AUTO_MAX_STATE = 1 << 23               # elements of y0 above which "auto" stays eager
_AUTO_MAX_OUTPUT_BYTES = 1 << 30        # ... or bytes of ys


def mode_of(options, key="hip_graph"):
    """True / False / "auto" from the option. Absent: the environment's TSDE_HIP_GRAPH ("auto", the default; "1": record
    on first use like `True`; "0"). TSDE_HIP_GRAPH=0 is also the kill switch: it wins over an explicit option, so a
    deployment can rule graphs out without touching call sites."""
    import os
    env = os.environ.get("TSDE_HIP_GRAPH", "auto").strip().lower()
    if env not in ("0", "1", "auto", "false", "true", "off", "on", ""):
        raise ValueError(f"TSDE_HIP_GRAPH must be 0, 1 or auto, got {env!r}.")
    if env in ("0", "false", "off"):
        return False
    default = True if env in ("1", "true", "on") else "auto"
    value = options.get(key, default) if options is not None else default
    if value is True or value is False:
        return value
    if value is None or value == "auto":
        return "auto"
    raise ValueError(f"options[{key!r}] must be True, False or 'auto', got {value!r}.")


class _Seen:
    """Cache entry of a structure that was solved (eagerly) once: the next solve of it is captured. `independent`: the
    screened run saw the drift and the diffusion touch disjoint memory, so the capture may try them as parallel
    branches (and keep that form if it is faster AND reproduces the sequential one)."""

    def __init__(self, independent=False):
        self.independent = independent


class _Refused:
    """Cache entry of a structure that must stay eager (why: `reason`)."""

    def __init__(self, reason):
        self.reason = reason


class _TooMuchState(Exception):
    pass


_MAX_STATES_PER_STRUCTURE = 8


def auto_key(cache, structure, base):
    """The "auto" cache key of a solve with structural signature `structure` on SDE object `base` -- ("auto", Python-side
    state, process switches) + structure -- or None when this structure stays eager: the state cannot be fingerprinted,
    or the object has shown more than `_MAX_STATES_PER_STRUCTURE` different states for this one structure (a counter
    bumped in `f`, a list that grows, a per-batch `sde.ctx = new_tensor`). Such an object never reaches a replay; it
    would pay for the screen and the fingerprint walk on every solve, so after that many misses the structure itself is
    refused and later solves come straight back here and leave by the first test."""
    churn = ("auto-churn",) + structure
    if isinstance(cache.get(churn), _Refused):
        return None
    state = python_state(base)
    if state is None:
        return None
    sig = ("auto", state, process_state()) + structure
    seen = cache.get(churn)
    if sig in cache:
        if seen is not None:
            seen.count = 0            # a state that came back: the object is not churning
        return sig
    if seen is None:
        seen = cache[churn] = _StatesSeen()
    # CONSECUTIVE misses only: a hyperparameter sweep that sets `sde.scale` eight times over the object's lifetime, with
    # solves that hit the cache in between, is eight legitimate states, not churn
    seen.count += 1
    if seen.count > _MAX_STATES_PER_STRUCTURE:
        # entries that never got as far as a replay go (they are the churn); a graph that has replayed stays usable
        for key in [k for k in cache if k[:1] == ("auto",) and k[3:] == structure]:
            if getattr(cache[key], "replays", 0) == 0:
                del cache[key]
        cache[churn] = _Refused(f"the SDE object's Python-side state differed on each of {seen.count - 1} consecutive solves "
                                "of this structure (a counter, a growing list, a fresh tensor attribute per call?)")
        return None
    return sig


class _StatesSeen:
    """Cache entry counting the CONSECUTIVE cache misses of one structure (see `auto_key`); any hit resets it."""
    count = 0


def due_for_a_check(replays):
    """Is replay number `replays` (1-based) of an accepted graph one that runs next to the eager path and is compared
    with it? The first two (probation), then on a geometric schedule -- 8, 64, 512, ... -- for as long as the graph
    lives: state the fingerprint cannot see (a C extension's global, an environment variable read in `f`) is caught
    late rather than never, at an amortised cost under 1/7 of a solve per 8."""
    if replays <= 2:
        return True
    while replays % 8 == 0:
        replays //= 8
    return replays == 1


_SIMPLE = (bool, int, float, complex, str, bytes, type(None), torch.dtype, torch.device, torch.Size)
# classes whose attributes are library code, not the caller's state
_LIBRARY_MODULES = ("torch", "builtins", "numpy", "collections", "abc", "typing", "functools")


def _is_library(x):
    module = getattr(x, "__module__", None) or ""
    return module.split(".")[0] in _LIBRARY_MODULES or module.startswith("torchsde_amd")


def _simple(x):
    """The entry of a plain value: with its type (1, 1.0 and True are different programs) and, for floats, by `repr`
    (nan equals itself, -0.0 is not 0.0). Compared by EQUALITY as part of a tuple -- never reduced to `hash()`:
    CPython hashes collide on everyday values (hash(-1) == hash(-2), hash(-1.0) == hash(-2.0))."""
    if isinstance(x, (float, complex)):
        return (type(x).__name__, repr(x))
    return (type(x).__name__, x)


def call_counters(obj):
    """{attribute: {"f": n, "g": n[, "h": n]}} for the PURE CALL COUNTERS of an SDE object -- the `self._nfe += 1` of the reference's
    test problems (tests/problems.py:60-66, 92-98, 118-124): an int attribute that the object's own code only ever increments
    by a constant inside `f` / `g` / `h` (and may hand out through a getter that nothing else in the class calls). Such a value
    cannot reach the dynamics, so a route that calls `f` and `g` another number of times than the stepwise loop may ignore it
    in its state comparison AND put it to the value the stepwise loop would have left (solvers._integrate_recognised). Decided
    on the bytecode of every function of the object's user classes; anything it does not recognise (another Python version's
    opcodes, a read anywhere else, a write outside `__init__`) leaves the attribute out: it then counts as state, as before."""
    import dis
    import sys
    import types
    if sys.version_info[:2] != (3, 10) or not hasattr(obj, "__dict__"):
        return {}
    names = [k for k, v in vars(obj).items() if type(v) is int and isinstance(k, str) and not k.startswith("_tsde")]
    if not names:
        return {}
    memo = (type(obj), tuple(names))                  # (the answer is a property of the classes' code and of these names)
    if memo in _CALL_COUNTERS:
        return _CALL_COUNTERS[memo]
    if len(_CALL_COUNTERS) >= 64:
        _CALL_COUNTERS.clear()
    found = _CALL_COUNTERS[memo] = _call_counters_of(obj, names)
    return found


_CALL_COUNTERS = {}


def _call_counters_of(obj, names):
    import dis
    import types
    functions = []                                   # (name in the class, code)
    for klass in type(obj).__mro__:
        if klass is object or _is_library(klass):
            continue
        for fname, value in vars(klass).items():
            if isinstance(value, (staticmethod, classmethod)):
                value = value.__func__
            elif isinstance(value, property):
                value = value.fget
            if isinstance(value, types.FunctionType):
                functions.append((fname, value.__code__, value.__globals__))
    found = {}
    for attr in names:
        incs, getters, ok, owners = {}, set(), True, set()
        for fname, code, scope in functions:
            nested = [k for k in code.co_consts if isinstance(k, types.CodeType)]
            if any(attr in c.co_names for c in nested):
                ok = False                           # (a lambda / inner function that names it)
                break
            for helper in code.co_names:             # module-level helpers the function may call: they must not name it
                value = scope.get(helper)
                if isinstance(value, types.FunctionType) and not _is_library(value) and attr in value.__code__.co_names:
                    ok = False
            if not ok:
                break
            if attr not in code.co_names:
                continue
            ins = list(dis.get_instructions(code))
            first_arg = code.co_varnames[0] if code.co_argcount else None
            if [i.opname for i in ins] == ["LOAD_FAST", "LOAD_ATTR", "RETURN_VALUE"] and ins[1].argval == attr:
                getters.add(fname)
                continue
            k = 0
            while k < len(ins) and ok:
                i = ins[k]
                if i.argval == attr and i.opname in ("LOAD_ATTR", "STORE_ATTR", "DELETE_ATTR", "LOAD_METHOD"):
                    pattern = [x.opname for x in ins[k - 2:k + 5]] if k >= 2 else []
                    if (i.opname == "LOAD_ATTR" and pattern == ["LOAD_FAST", "DUP_TOP", "LOAD_ATTR", "LOAD_CONST", "INPLACE_ADD",
                                                                "ROT_TWO", "STORE_ATTR"]
                            and ins[k - 2].argval == first_arg and ins[k + 4].argval == attr and type(ins[k + 1].argval) is int
                            and fname in ("f", "g", "h")):      # (h: the prior drift, only called under logqp)
                        if (fname, id(code)) not in owners and any(n == fname for n, _ in owners):
                            ok = False               # (the same method counts in two classes of the MRO: which ones run?)
                            break
                        owners.add((fname, id(code)))
                        incs[fname] = incs.get(fname, 0) + ins[k + 1].argval
                        k += 5
                        continue
                    if i.opname == "STORE_ATTR" and fname == "__init__":
                        k += 1
                        continue
                    ok = False
                k += 1
            if not ok:
                break
        if ok and incs and not any(g in code.co_names for g in getters for _, code, _ in functions):
            found[attr] = incs
    return found


def python_state(obj, budget=4096, ignore=()):
    """A hashable fingerprint -- compared by equality, it IS the cache key -- of the Python-side state a recorded graph
    would bake in. For `obj` and everything reachable from it: plain attribute values; the identity (storage, shape,
    STRIDES, offset, dtype) of every tensor -- not tensor contents, which replays read live; flags such as `training`;
    plain class attributes along the MRO of user classes; and for every function or method found on the way (the
    SDE's `f`, `g`, helpers they name) the code object, default arguments, the contents of closure cells and the values
    of the module globals its code names -- a drift that reads a module-level `SCALE` or a closed-over coefficient is
    re-recorded when that number changes (the reference re-runs user code every step, base_solver.py:114-149, and sees
    the change at once). None when there is too much to fingerprint cheaply or something is unhashable: then "auto"
    stays eager."""
    import functools
    import types
    out, seen = [], set()
    left = [budget]

    def function(fn, depth):
        """A Python function: its code, defaults, closure contents and the globals its code (and nested code) names."""
        if id(fn) in seen:
            out.append(("F", id(fn)))
            return
        seen.add(id(fn))
        code = fn.__code__
        out.append(("F", id(code)))
        if fn.__defaults__:
            walk(fn.__defaults__, depth + 1)
        if fn.__kwdefaults__:
            walk(fn.__kwdefaults__, depth + 1)
        for cell in fn.__closure__ or ():
            try:
                walk(cell.cell_contents, depth + 1)
            except ValueError:           # an empty cell
                out.append(("empty-cell",))
        names, stack = [], [code]
        while stack:
            c = stack.pop()
            names.extend(c.co_names)
            stack.extend(k for k in c.co_consts if isinstance(k, types.CodeType))
        scope = fn.__globals__
        for name in dict.fromkeys(names):            # (first occurrence order, no duplicates)
            if name in scope:
                value = scope[name]
                out.append(("G", name))
                if isinstance(value, (types.ModuleType, type)) or (callable(value) and _is_library(value)):
                    out.append(("O", id(value)))
                else:
                    walk(value, depth + 1)

    def class_attributes(cls, depth):
        """Plain values, tensors and functions in the class bodies of the user's classes along the MRO."""
        for klass in cls.__mro__:
            if klass is object or _is_library(klass) or id(klass) in seen:
                continue
            seen.add(id(klass))
            out.append(("C", klass.__qualname__))
            for name, value in vars(klass).items():
                if name.startswith("__") and name.endswith("__") and not callable(value):
                    continue
                if isinstance(value, (staticmethod, classmethod)):
                    value = value.__func__
                elif isinstance(value, property):
                    value = value.fget
                if isinstance(value, _SIMPLE) or torch.is_tensor(value) or isinstance(value, (types.FunctionType, list, tuple, dict)):
                    out.append(("A", name))
                    walk(value, depth + 1)

    def walk(x, depth):
        left[0] -= 1
        if left[0] < 0:
            raise _TooMuchState
        if isinstance(x, _SIMPLE):
            out.append(_simple(x))
        elif torch.is_tensor(x):
            try:
                where = (x.data_ptr(), tuple(x.stride()), x.storage_offset())
            except Exception:           # sparse / nested layouts have no single data pointer: identity of the object
                where = ("O", id(x))
            out.append(("T", where, tuple(x.shape), x.dtype, x.requires_grad))
        elif id(x) in seen or depth > 8:
            out.append(("O", id(x)))
        elif isinstance(x, (list, tuple, set, frozenset)):
            seen.add(id(x))
            out.append((type(x).__name__, len(x)))
            for e in x:
                walk(e, depth + 1)
        elif isinstance(x, dict):
            seen.add(id(x))
            # (this package's own caches on the object -- `_tsde_*` attributes -- are not its state)
            items = [(k, v) for k, v in x.items() if not (isinstance(k, str) and k.startswith("_tsde"))]
            out.append(("dict", len(items)))
            for k, v in items:
                out.append(_simple(k) if isinstance(k, _SIMPLE) else ("O", id(k)))
                walk(v, depth + 1)
        elif isinstance(x, types.FunctionType):
            function(x, depth)
        elif isinstance(x, types.MethodType):
            out.append(("M",))
            walk(x.__func__, depth + 1)
            walk(x.__self__, depth + 1)
        elif isinstance(x, functools.partial):
            out.append(("P",))
            walk(x.func, depth + 1)
            walk(x.args, depth + 1)
            walk(x.keywords, depth + 1)
        elif isinstance(x, (types.ModuleType, type, types.BuiltinFunctionType)):
            out.append(("O", id(x)))
        elif isinstance(x, torch.nn.Module) or (hasattr(x, "__dict__") and not _is_library(type(x))):
            seen.add(id(x))
            out.append((type(x).__qualname__,))
            class_attributes(type(x), depth)
            own = vars(x)
            if ignore and x is obj:                  # (pure call counters: see `call_counters`)
                own = {k: v for k, v in own.items() if k not in ignore}
            walk(own, depth + 1)
            call = getattr(type(x), "__call__", None)
            if isinstance(call, types.FunctionType) and not _is_library(call):
                function(call, depth + 1)
        else:
            out.append(("O", id(x)))     # generators, foreign objects: identity only

    try:
        walk(obj, 0)
        key = tuple(out)
        hash(key)
    except (_TooMuchState, TypeError, RecursionError):
        return None
    return key


def process_state():
    """Process-wide switches that change what the same torch code computes: part of every "auto" cache key."""
    return (torch.is_autocast_enabled(), str(torch.get_autocast_dtype("cuda")), torch.get_float32_matmul_precision(),
            torch.backends.cuda.matmul.allow_tf32, torch.are_deterministic_algorithms_enabled(),
            str(torch.get_default_dtype()))


def _same_tensors(xs, ys, exact=True):
    """Do two lists of tensors agree (NaNs in the same places count as agreement)? One host sync."""
    ok = True
    for a, b in zip(xs, ys):
        if a.shape != b.shape:
            return False
        if exact:
            ok = ok & ((a == b) | (a.isnan() & b.isnan())).all()
        else:
            scale = torch.nan_to_num(b.abs()).max().clamp_min(1e-30)
            ok = ok & ((torch.nan_to_num(a - b).abs().max() <= 1e-3 * scale) & (a.isnan() == b.isnan()).all())
    return bool(ok)


# Operators a drift / diffusion may use and still be recorded silently: what elementwise networks are made of. Anything
# else (factorisations, FFT plans, sorting, random numbers, custom extension ops ...) MAY be capture-safe, but a capture
# that fails poisons the HIP context of the whole process on ROCm 7.2 (every later call reports
# hipErrorStreamCaptureInvalidated), so "auto" only records code it has SEEN to consist of these. (base names of
# `aten::` operators; in-place and `.out` variants are covered by stripping the trailing underscore / overload.)
_CAPTURE_SAFE = frozenset("""
add sub rsub mul div true_divide floor_divide neg abs absolute exp exp2 expm1 log log1p log2 log10 sqrt rsqrt pow square
reciprocal sin cos tan sinh cosh tanh asin acos atan atan2 asinh acosh atanh sigmoid silu relu relu6 gelu elu selu celu
leaky_relu prelu softplus hardtanh hardsigmoid hardswish mish log_sigmoid log_sigmoid_forward logit clamp clamp_min
clamp_max clip minimum maximum fmin fmax where sign sgn floor ceil round trunc frac fmod remainder lerp addcmul addcdiv erf
erfc erfinv lgamma digamma nan_to_num threshold softshrink hardshrink logical_and logical_or logical_not logical_xor eq ne lt
le gt ge isnan isinf isfinite isneginf isposinf bitwise_and bitwise_or bitwise_xor bitwise_not heaviside hypot xlogy
copy fill zero _to_copy to type_as positive conj real imag
empty empty_like empty_strided zeros zeros_like ones ones_like full full_like new_empty new_zeros new_ones new_full
new_empty_strided arange linspace scalar_tensor lift_fresh lift clone contiguous detach alias eye
view _unsafe_view reshape _reshape_alias expand expand_as permute transpose t squeeze unsqueeze slice select narrow split
split_with_sizes chunk unbind unfold as_strided diagonal flatten unflatten view_as movedim swapaxes swapdims numpy_T mT mH
unsafe_split unsafe_chunk unsafe_split_with_sizes tensor_split hsplit vsplit
mm bmm addmm baddbmm matmul linear mv addmv dot vdot outer ger addr addbmm einsum tensordot bilinear _addmm_activation
sum mean prod amax amin max min norm linalg_vector_norm var std var_mean std_mean logsumexp softmax _softmax log_softmax
_log_softmax cumsum cumprod logcumsumexp argmax argmin all any count_nonzero nansum nanmean
cat concat concatenate stack hstack vstack repeat tile flip roll index_select gather scatter scatter_add scatter_reduce
index_add index_copy index_fill masked_fill diag_embed tril triu diag broadcast_to broadcast_tensors expand_copy constant_pad_nd pad
layer_norm native_layer_norm group_norm native_group_norm rms_norm _fused_rms_norm
softmax_backward_data _softmax_backward_data tanh_backward sigmoid_backward softplus_backward silu_backward gelu_backward
elu_backward threshold_backward leaky_relu_backward hardtanh_backward native_layer_norm_backward sum_to_size
""".split())


_RECORDER = None      # the recorder of the screened run in progress (sde.ForwardSDE reports its drift / diffusion phases)


def _storages(obj, out):
    """Data pointers of every tensor in a (nested) argument / result structure."""
    if torch.is_tensor(obj):
        try:
            out.add(obj.untyped_storage().data_ptr())
        except Exception:       # meta / fake tensors: nothing to alias
            pass
    elif isinstance(obj, (list, tuple)):
        for x in obj:
            _storages(x, out)
    elif isinstance(obj, dict):
        for x in obj.values():
            _storages(x, out)


class _OperatorRecorder(TorchDispatchMode):
    """Watches a screened eager run: notes every operator outside `_CAPTURE_SAFE`, and -- for the first few drift /
    diffusion evaluations, whose phases `sde.ForwardSDE._f_then_g` announces -- which storages each phase reads and
    writes. Drift and diffusion are INDEPENDENT if neither touches what the other writes (a diffusion that reuses a
    tensor the drift computed and cached, a shared scratch buffer ...): only then may they be recorded as parallel
    branches of a graph. Results of a tracked pair are kept alive until the pair ends, so that the allocator cannot hand
    the drift's freed temporaries to the diffusion and fake a dependence."""

    TRACKED_PAIRS = 4

    def __init__(self):
        super().__init__()
        self.unknown = set()
        self.phase = None
        self.pairs = 0
        self.shared = False
        self.touched = {"f": set(), "g": set()}
        self.written = {"f": set(), "g": set()}
        self.keep = []

    def enter_phase(self, phase):
        """phase: "f", "g" or None (the pair is over: it is judged by itself, and its tensors are let go)."""
        if phase is None:
            if (self.written["f"] & self.touched["g"]) or (self.written["g"] & self.touched["f"]):
                self.shared = True
            for table in (self.touched, self.written):
                table["f"].clear()
                table["g"].clear()
            self.keep.clear()
            self.pairs += 1
        self.phase = phase if self.pairs < self.TRACKED_PAIRS else None

    def independent(self):
        """Did at least one drift / diffusion pair run, and none with a storage written by one and touched by the other?"""
        return self.pairs > 0 and not self.shared

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = getattr(getattr(func, "_schema", None), "name", str(func))
        namespace, _, base = name.partition("::")
        if namespace != "aten" or (base.rstrip("_") not in _CAPTURE_SAFE and base not in _CAPTURE_SAFE):
            self.unknown.add(name)
        result = func(*args, **(kwargs or {}))
        phase = self.phase
        if phase is not None:
            ins, outs = set(), set()
            _storages(args, ins)
            _storages(kwargs, ins)
            _storages(result, outs)
            self.touched[phase] |= ins | outs
            # what an operator returns in memory that was not among its inputs it has written (fresh); what it returns
            # in an input's memory is a view -- unless the schema says the operator mutates (in place, out=)
            self.written[phase] |= (outs - ins)
            if getattr(getattr(func, "_schema", None), "is_mutable", False):
                self.written[phase] |= (outs & ins)
            self.keep.append(result)
        return result


def replays_are_stable(replay, outputs, disturb=None, extra_replays=2):
    """Does replaying a freshly recorded graph keep giving what its first replay gave -- also after other work has run
    on the device in between? It should, trivially. But on this stack (ROCm 7.2, torch 2.10) a graph that holds several
    multi-block torch reductions (`x.sum(0)` over a few thousand rows: the parameter gradients of a broadcast `w * y`)
    is right when first replayed and wrong, stably, once an eager reduction and a host synchronisation have come between
    two replays (tools/probe_graph_reduction4.py, profiles/r3j_probe_graph_reduction.txt: twenty column sums of a
    4096 x 128 tensor in one graph are 23 % off from the second replay on; the backward sweep of `sdeint_adjoint` at
    B = 4096, d = 128 returned inf for per-channel parameters). The nodes at fault are the MEMSET nodes with which ATen
    zeroes the semaphores of such reductions, and `_capturing` rewrites them as kernel nodes, which cures every case
    found (tools/probe_graph_surgery.py, profiles/r3q_probe_graph_surgery.txt; the runtime flag
    DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 does too). This check stays as the second line: every recorded graph is verified
    before it is trusted: `replay()` runs the graph, `outputs()` lists its result tensors, `disturb()` runs the same
    computation eagerly (allocations, reductions and all: what a caller does between two solves) before each further
    replay. NaN in the same place counts as equal."""
    first = [o.clone() for o in outputs()]
    for _ in range(extra_replays):
        if disturb is not None:
            disturb()
        replay()
        if not _same_tensors(outputs(), first, exact=True):
            return False
    return True


_UNSTABLE = ("replaying the recorded graph does not reproduce its own first replay (a known fault of recorded memset "
             "nodes on this runtime: graph.replays_are_stable)")


def run_screened(fn, verdict=None):
    """`fn()` -- the launch-only part of an eager solve -- watched for everything that would make recording it unsafe:
    host synchronisation (torch's sync-debug mode) and operators outside `_CAPTURE_SAFE` (a dispatch-mode recorder).
    Returns (result, reason): `reason` is None if the code may be recorded. Other warnings raised meanwhile are re-issued.
    `verdict`: a dict that receives `independent` (may drift and diffusion be recorded as parallel branches?)."""
    global _RECORDER
    previous = torch.cuda.get_sync_debug_mode()
    recorder = _OperatorRecorder()
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        torch.cuda.set_sync_debug_mode("warn")
        _RECORDER = recorder
        try:
            with recorder:
                result = fn()
        finally:
            _RECORDER = None
            torch.cuda.set_sync_debug_mode(previous)
    if verdict is not None:
        verdict["independent"] = recorder.independent()
    recorder.keep.clear()
    reason = None
    for w in caught:
        text = str(w.message)
        if "called a synchronizing" in text:          # c10's "called a synchronizing CUDA operation"
            reason = "the code synchronises with the host"
        elif "Synchronization debug mode is a prototype feature" not in text:    # (torch's one-time notice about the mode)
            warnings.warn_explicit(w.message, w.category, w.filename, w.lineno)
    if reason is None and recorder.unknown:
        reason = "operators that are not known to be capture-safe: " + ", ".join(sorted(recorder.unknown)[:6])
    return result, reason


def _auto_eligible(bm, y0, n_out):
    return (y0.is_cuda and isinstance(bm, BrownianInterval) and bm._rootW is None and bm._rootH is None
            and 0 < y0.numel() <= _AUTO_MAX_STATE and n_out * y0.numel() * y0.element_size() <= _AUTO_MAX_OUTPUT_BYTES
            and not torch.cuda.is_current_stream_capturing())


@contextlib.contextmanager
def _drift_then_diffusion(sde):
    """Inside: the user's drift and diffusion are recorded one after the other (no parallel graph branches)."""
    had = getattr(sde, "overlap_f_g", None)
    if had:
        sde.overlap_f_g = False
    try:
        yield
    finally:
        if had:
            sde.overlap_f_g = True


def auto_solve(solver, y0, ts, extra0=()):
    """The "auto" route of a forward solve without autograd: None -> the caller runs it eagerly; else (ys, extras)."""
    bm = solver.bm
    from . import timegrid
    ts_host = timegrid.ts_to_host(ts)
    if not _auto_eligible(bm, y0, len(ts_host)):
        return None
    _, base = _wrapper_chain(solver.sde)
    cache = _cache_of(base)
    if not bm.frozen:
        bm.adopt_grid(timegrid.build(ts_host, solver.dt).t_f64())
    sig = auto_key(cache, _signature(solver, y0, ts_host), base)
    if sig is None:
        return None
    entry = cache.get(sig)
    if entry is None:
        # first solve of this structure: eager, and watched -- code that synchronises with the host cannot be captured
        plan = solver._plan(y0, ts)
        solver._extra = tuple(extra0)
        verdict = {}
        try:
            ys, reason = run_screened(lambda: solver._run(plan, y0), verdict)
        except Exception as e:
            # whatever went wrong under the screen (the user's own error, or code that does not run under a dispatch
            # mode): this structure stays eager, and the plain eager run that follows reports the error if it is real
            _remember(cache, sig, _Refused(f"the screened run raised {type(e).__name__}"))
            solver._extra = tuple(extra0)
            return None
        _remember(cache, sig, _Seen(verdict["independent"]) if reason is None else _Refused(reason))
        return ys, solver._extra
    if isinstance(entry, _Refused):
        return None
    if isinstance(entry, _Seen):
        try:
            if entry.independent:
                captured = faster_of_sequential_and_parallel(
                    solver.sde if hasattr(solver.sde, "_f_then_g") else None,
                    lambda: _CapturedSolve(solver, y0, ts, extra0, verify=True), y0.device)
            else:
                with _drift_then_diffusion(solver.sde):
                    captured = _CapturedSolve(solver, y0, ts, extra0, verify=True)
        except Exception as e:     # (a Python-level failure inside the recording; out of memory for the second pool)
            cache[sig] = _Refused(f"capture failed: {type(e).__name__}: {e}")
            solver._extra = tuple(extra0)
            return None
        if not captured.verified:
            cache[sig] = _Refused(_UNSTABLE if not captured.stable else "the recorded graph did not reproduce the eager solve")
            return captured.eager_result
        _remember(cache, sig, captured.accept())
        return captured.result()
    return _replay_on_probation(entry, cache, sig, solver, bm, y0, ts, extra0)


def _replay_ms(graphs, device, max_rounds=4):
    """Duration of one replay of each of `graphs`: the best of up to `max_rounds` replays each, taken in alternation
    after one untimed replay (so that clocks, caches and allocator state are the same for all of them), events on the
    current stream. Stops after the second round when the candidates are more than 10 % apart by then (one round can
    hold an outlier)."""
    for g in graphs:
        g.replay()
    best = [float("inf")] * len(graphs)
    for round_ in range(max_rounds):
        for i, g in enumerate(graphs):
            start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            start.record()
            g.replay()
            stop.record()
            stop.synchronize()
            best[i] = min(best[i], start.elapsed_time(stop))
        if round_ >= 1 and max(best) > 1.1 * min(best):
            break
    return best


def faster_of_sequential_and_parallel(forward_sde, capture, device):
    """Capture a solve (or a backward sweep) twice -- the user's drift and diffusion recorded one after the other, and
    as parallel branches of the graph (sde.ForwardSDE._f_beside_g) -- replay both, keep the faster one.

    Whether the parallel form pays depends on the user's code: two branches of several kernels each (perceptron drift
    and diffusion: -18 % on BASELINE configs[2]) overlap their ramp-up and drain, while for one short kernel per branch
    the graph's fork / join edges cost more than they hide (+30 % on the headline's `mu * y`, `sigma * y`). It is
    measured once per cached graph, on the first solve with that structure. `capture()` -> an object with `.graph`."""
    tunable = (forward_sde is not None and getattr(forward_sde, "overlap_f_g", False)
               and getattr(forward_sde.f_and_g, "__func__", None) is type(forward_sde)._f_then_g)
    if not tunable:
        return capture()
    try:
        forward_sde.overlap_f_g = False
        reserved = torch.cuda.memory_reserved(device)
        sequential = capture()
        if sequential is None or getattr(sequential, "verified", True) is False or not getattr(sequential, "stable", True):
            return sequential
        # the second graph needs a memory pool of its own until the loser is dropped: no tuning when that does not
        # comfortably fit (a solve that fitted before must not run out of memory because of the tuner)
        pool = max(torch.cuda.memory_reserved(device) - reserved, 0)
        if torch.cuda.mem_get_info(device)[0] < 2 * pool:
            sequential.tuning = {"kept": "sequential", "why": "not enough free memory to try the parallel form"}
            return sequential
        forward_sde.overlap_f_g = True
        parallel = capture()
    finally:
        forward_sde.overlap_f_g = True
    if parallel is None or getattr(parallel, "verified", True) is False or not getattr(parallel, "stable", True):
        return sequential
    t_seq, t_par = _replay_ms([sequential.graph, parallel.graph], device)
    # the parallel form is only an option if it computes the same thing: drift and diffusion code that shares buffers,
    # caches or a random generator gives other values when its two halves run side by side
    agree = _same_tensors(parallel.outputs(), sequential.outputs(), exact=parallel.exact_outputs)
    tuning = {"sequential_ms": t_seq, "parallel_ms": t_par, "parallel_agrees": agree}
    if agree and t_par < 0.9 * t_seq:       # (the parallel form has to earn its second stream, beyond timing noise)
        sequential = None              # (drops the losing graph and its memory pool now, not at the caller's return)
        keep, tuning["kept"] = parallel, "parallel"
    else:
        parallel = None
        keep, tuning["kept"] = sequential, "sequential"
    keep.tuning = tuning
    if not keep.still_right():      # (the timing replays ran next to another graph: the fault of replays_are_stable)
        keep.stable = False
        if hasattr(keep, "verified"):
            keep.verified = False
    return keep


def new_graph():
    """A torch CUDAGraph that keeps its hipGraph_t after the capture and is instantiated by its first replay, so that
    `_capturing` can rewrite its memset nodes in between."""
    try:
        graph = torch.cuda.CUDAGraph(keep_graph=True)
    except TypeError:       # a torch without `keep_graph` (< 2.8): no rewriting, the replay checks alone stand guard
        return torch.cuda.CUDAGraph()
    graph.memset_nodes = None
    return graph


def _memset_nodes_to_kernels(graph):
    """Recorded MEMSET nodes -- ATen's multi-block reductions zero their semaphores with `hipMemsetAsync`, so a drift
    with `y.mean(0)` in it has them, and so has every backward sweep whose parameter gradients are column sums over
    more rows than one block reduces -- stop doing their work on this runtime once an eager memset and a host
    synchronisation have come between two replays (tools/probe_graph_reduction4.py; the fault `replays_are_stable`
    looks for). Kernel nodes do not have the problem: `tsde_graph_memset_nodes_to_kernels` (csrc/graph_nodes.hip) swaps
    each memset node for a fill kernel with the same edges before the graph is instantiated."""
    import ctypes
    from . import _native
    found, replaced = ctypes.c_int(0), ctypes.c_int(0)
    _native.check(_native.load().tsde_graph_memset_nodes_to_kernels(
        ctypes.c_void_p(graph.raw_cuda_graph()), ctypes.byref(found), ctypes.byref(replaced)),
        "tsde_graph_memset_nodes_to_kernels")
    graph.memset_nodes = (found.value, replaced.value)


@contextlib.contextmanager
def _capturing(graph, device, **kwargs):
    """`torch.cuda.graph(graph, capture_error_mode="thread_local")` with the cyclic GC paused -- and the caller's stream
    restored when the capture fails: `torch.cuda.graph.__exit__` raises from `capture_end()` BEFORE it leaves its stream
    context, which would leave every later launch of the process on the dead capture stream. Graphs from `new_graph()`
    have their memset nodes rewritten as kernels once the capture has ended."""
    previous = torch.cuda.current_stream(device)
    try:
        with _no_gc(), torch.cuda.graph(graph, capture_error_mode="thread_local", **kwargs):
            yield
    except BaseException:
        torch.cuda.set_stream(previous)
        raise
    if hasattr(graph, "memset_nodes") and _REWRITE_MEMSET_NODES:
        _memset_nodes_to_kernels(graph)


_REWRITE_MEMSET_NODES = True     # (tests and tools/probe_graph_surgery.py switch it off to show the fault)


def _replay_on_probation(captured, cache, sig, solver, bm, y0, ts, extra0):
    """Replay a recorded forward solve; while it is on probation (its first replays in real use) also run the solve
    eagerly and compare: the fault `replays_are_stable` looks for shows only after other work has run on the device. A
    graph that fails is dropped for good and the eager result returned."""
    out = captured.replay(bm, y0, extra0)
    captured.replays += 1
    # probation for every graph; the geometric schedule only for the silent default (`True`: the caller vouches)
    if captured.replays <= 2 or (sig[:1] == ("auto",) and due_for_a_check(captured.replays)):
        solver._extra = tuple(extra0)
        eager = solver._run(solver._plan(y0, ts), y0)
        eager_extra = tuple(solver._extra)
        if not _same_tensors([out[0]] + list(out[1]), [eager] + list(eager_extra)):
            cache[sig] = _Refused(_UNSTABLE)
            return eager, eager_extra
    return out


class _CapturedSolve:
    exact_outputs = True
    replays = 0               # replays in real use so far: `due_for_a_check` says which of them run beside the eager path

    def outputs(self):
        return [self.ys] + list(self.extra_out)

    def __init__(self, solver, y0, ts, extra0=(), verify=False):
        bm = solver.bm
        device = y0.device
        self.seed_dev = torch.zeros(1, dtype=torch.int64, device=device)
        self.y_in = torch.empty_like(y0, memory_format=torch.contiguous_format)
        self.y_in.copy_(y0)
        # solvers that carry state between steps (reversible Heun: f, g, z) get it as further static inputs
        self.extra_in = [torch.empty_like(e, memory_format=torch.contiguous_format).copy_(e) for e in extra0]
        self._set_seed(bm)
        bm._entropy_dev = self.seed_dev
        # the captured kernels hold raw pointers into this Brownian motion (device copy of the cell edges) and into
        # the plan's stage-time tensors: keep both alive for as long as the graph. (Not the solver: it references the
        # SDE object that owns this cache, and a reference cycle would leave the graph's destruction to the GC.)
        self._keepalive = bm
        try:
            self.plan = solver._plan(self.y_in, ts)
            side = torch.cuda.Stream(device=device)
            side.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(side):           # warm-up outside capture (lazy inits, allocator)
                solver._extra = tuple(self.extra_in)
                warm = solver._run(self.plan, self.y_in)
                warm_extra = tuple(solver._extra)
            torch.cuda.current_stream(device).wait_stream(side)
            if not verify:
                del warm, warm_extra
            self.graph = new_graph()
            # thread_local: API calls from other threads (e.g. the RCCL watchdog of a multi-GPU run) must not
            # invalidate this capture
            with _capturing(self.graph, device):
                solver._extra = tuple(self.extra_in)
                self.ys = solver._run(self.plan, self.y_in)
                self.extra_out = tuple(solver._extra)
        finally:
            bm._entropy_dev = None
        self.graph.replay()     # capture only records: run once so that `ys` holds this solve's result

        def eagerly():
            solver._extra = tuple(self.extra_in)
            solver._run(self.plan, self.y_in)
        bm._entropy_dev = self.seed_dev
        try:
            self.stable = replays_are_stable(self.graph.replay, self.outputs, eagerly)
        finally:
            bm._entropy_dev = None
        # what the outputs must (still) be after any further replay on these inputs -- `still_right`, asked again after
        # the tuner has replayed this graph next to another one: the eager solve ("auto"), else the first replay
        self.reference = [warm] + list(warm_extra) if verify else [o.clone() for o in self.outputs()]
        if verify:              # "auto": the replay must be the eager solve it was recorded beside, bit for bit
            self.verified = self.stable and self.still_right()
            self.eager_result = (warm, warm_extra)

    def still_right(self):
        return _same_tensors(self.outputs(), self.reference, exact=True)

    def accept(self):
        """Checks passed: let go of the reference copies."""
        self.reference = None
        self.eager_result = None
        return self

    def _set_seed(self, bm):
        key = bm._key
        self.seed_dev.fill_(key - (1 << 64) if key >= (1 << 63) else key)   # two's complement into int64

    def result(self):
        return self.ys.clone(), tuple(e.clone() for e in self.extra_out)

    def replay(self, bm, y0, extra0=()):
        self.y_in.copy_(y0)
        for dst, src in zip(self.extra_in, extra0):
            dst.copy_(src)
        self._set_seed(bm)
        self.graph.replay()
        return self.result()


def _wrapper_chain(sde):
    """What the per-call wrappers around the user's SDE object do: (wrapper class, renamed methods) from the outside
    in. Two calls on the same object with different `names=` (RenameMethodsSDE) or `logqp=` (SDELogqp) run different
    drift / diffusion code, so they must not share a captured graph."""
    chain = []
    while hasattr(sde, "_base_sde"):
        # (`getattr(v, "__name__", repr(v))` would evaluate the repr -- of a bound method: the whole module's -- at every solve)
        renamed = tuple(sorted((k, v.__name__ if hasattr(v, "__name__") else repr(v)) for k, v in vars(sde).items()
                               if k in ("f", "g", "h", "g_prod", "f_and_g", "f_and_g_prod") and callable(v)))
        chain.append((type(sde).__name__, renamed))
        sde = sde._base_sde
    return tuple(chain), sde


def _signature(solver, y0, ts_host):
    bm = solver.bm
    chain, base = _wrapper_chain(solver.sde)
    # the captured kernels read parameter STORAGE: a re-bound or moved parameter (`.to()`, `p.data = ...`) is another graph
    params = tuple(p.data_ptr() for p in base.parameters()) if hasattr(base, "parameters") else ()
    return (type(solver).__name__, chain, getattr(solver.sde, "sde_type", None), getattr(solver.sde, "noise_type", None),
            params, tuple(y0.shape), y0.dtype, str(y0.device), tuple(ts_host.tolist()),
            float(solver.dt), tuple(bm.shape), bm.levy_area_approximation, bm.row_offset,
            None if bm._edges is None else bm._edges.tobytes(), bm._max_depth, bm._snap,
            tuple(sorted((k, v) for k, v in solver.options.items() if isinstance(v, (bool, int, float, str)))))


def replay_or_capture(solver, y0, ts, extra0=()):
    """Run the solve through a cached HIP graph (capturing it on first use); returns (ys, extra solver state)."""
    bm = solver.bm
    if not isinstance(bm, BrownianInterval) or bm._rootW is not None or bm._rootH is not None:
        warnings.warn("hip_graph=True needs a torchsde_amd.BrownianInterval without pinned W/H; running eagerly.")
        solver._extra = tuple(extra0)
        return solver._run(solver._plan(y0, ts), y0), solver._extra
    from . import timegrid
    ts_host = timegrid.ts_to_host(ts)
    base = solver.sde
    while hasattr(base, "_base_sde"):    # ForwardSDE / RenameMethodsSDE / SDELogqp wrappers are rebuilt per call
        base = base._base_sde
    cache = _cache_of(base)
    # the grid the Brownian motion will have after adoption is part of the signature
    probe_plan_needed = not bm.frozen
    if probe_plan_needed:
        bm.adopt_grid(timegrid.build(ts_host, solver.dt).t_f64())
    sig = _signature(solver, y0, ts_host)
    captured = cache.get(sig)
    if captured is None:
        captured = faster_of_sequential_and_parallel(solver.sde if hasattr(solver.sde, "_f_then_g") else None,
                                                     lambda: _CapturedSolve(solver, y0, ts, extra0), y0.device)
        if not captured.stable:
            warnings.warn(f"hip_graph=True: {_UNSTABLE}; running eagerly.")
            captured = _Refused(_UNSTABLE)
        else:
            captured.accept()
        _remember(cache, sig, captured)
        if isinstance(captured, _Refused):
            solver._extra = tuple(extra0)
            return solver._run(solver._plan(y0, ts), y0), solver._extra
        return captured.result()
    if isinstance(captured, _Refused):
        solver._extra = tuple(extra0)
        return solver._run(solver._plan(y0, ts), y0), solver._extra
    return _replay_on_probation(captured, cache, sig, solver, bm, y0, ts, extra0)


class _CapturedBackward:
    """The launch-only backward sweep of ``sdeint_adjoint`` (``adjoint._run_backward``: re-materialised increments,
    the user's f/g and their VJPs through autograd, ``tsde_aug_update``) as ONE HIP graph. Static inputs: the stored
    forward states ``ys`` and the incoming gradients ``grad_ys`` -- plus, for the reversible-Heun pair, the solver's
    final (f, g, z) and their cotangents -- (copied in before each replay), the Brownian seed
    (device word) and the parameters themselves (read in place: an optimiser step is seen by the next replay)."""

    exact_outputs = False     # autograd orders the sums of a recorded sweep by per-thread sequence numbers
    replays = 0               # `due_for_a_check`: which replays in real use are compared with the eager sweep

    def outputs(self):
        return list(self.out)

    def __init__(self, run, bm, inputs, keepalive=(), verify=False):
        device = inputs[0].device
        self.seed_dev = torch.zeros(1, dtype=torch.int64, device=device)
        self.static = [torch.empty_like(x, memory_format=torch.contiguous_format) for x in inputs]
        self._load(bm, inputs)
        bm._entropy_dev = self.seed_dev
        # the captured kernels point into the plan (stage-time tensors) and into the Brownian motion (device copy of
        # the cell edges): keep those alive for as long as the graph (but not `run`, which references the SDE object
        # that owns this cache)
        self._keepalive = (bm,) + tuple(keepalive)
        try:
            side = torch.cuda.Stream(device=device)
            side.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(side):           # warm-up outside capture (lazy inits, allocator)
                warm = [o.clone() for o in run(*self.static)]
            torch.cuda.current_stream(device).wait_stream(side)
            if not verify:
                del warm
            self.graph = new_graph()
            with _capturing(self.graph, device):
                self.out = list(run(*self.static))
            # a first replay, then the check that later ones repeat it (replays_are_stable); "auto" also wants the first
            # one to give the eager sweep's gradients
            self._load(bm, inputs)
            self.graph.replay()
            self.stable = replays_are_stable(self.graph.replay, self.outputs, lambda: run(*self.static))
            self.reference = warm if verify else [o.clone() for o in self.out]
            self.exact_reference = not verify      # (eager sweep: autograd's summation order; own first replay: exact)
            if verify:
                self.verified = self.stable and self.still_right()
        finally:
            bm._entropy_dev = None

    def still_right(self):
        return _same_tensors(self.out, self.reference, exact=self.exact_reference)

    def accept(self):
        self.reference = None
        return self

    def _load(self, bm, inputs):
        for dst, src in zip(self.static, inputs):
            dst.copy_(src)
        key = bm._key
        self.seed_dev.fill_(key - (1 << 64) if key >= (1 << 63) else key)

    def replay(self, bm, inputs):
        self._load(bm, inputs)
        self.graph.replay()
        return [o.clone() for o in self.out]


def cached_backward(sde, bm, signature, capture, auto=False, tuned_capture=None):
    """The cached HIP graph of the adjoint's backward sweep for this structure; `capture()` builds it on a miss
    (and may return None: then nothing is cached and the backward pass runs eagerly).
    `signature` identifies the sweep's structure; the Brownian structure is appended here.
    `auto`: the "auto" rules of this module (`tuned_capture`: the variant that also tries drift and diffusion as
    parallel branches, used when the screened sweep found them independent) -- returns (graph or None, watch) where
    `watch`, if not None, is a callable
    the eager backward pass must report to (`watch(reason or None)`, see `run_screened`) so that the NEXT call knows
    whether to capture."""
    if bm._rootW is not None or bm._rootH is not None:
        if not auto:
            warnings.warn("hip_graph=True needs a torchsde_amd.BrownianInterval without pinned W/H; running eagerly.")
        return (None, None) if auto else None
    base = sde
    while hasattr(base, "_base_sde"):
        base = base._base_sde
    cache = _cache_of(base)
    sig = signature + (tuple(bm.shape), bm.levy_area_approximation, bm.row_offset,
                       None if bm._edges is None else bm._edges.tobytes(), bm._max_depth, bm._snap)
    if not auto:
        captured = cache.get(sig)
        if captured is None:
            captured = capture()
            if captured is not None and not captured.stable:
                warnings.warn(f"adjoint_options['hip_graph']=True: {_UNSTABLE}; the backward pass runs eagerly.")
                captured = _Refused(_UNSTABLE)
            elif captured is not None:
                captured.accept()
            if captured is not None:
                _remember(cache, sig, captured)
        return None if isinstance(captured, _Refused) else captured
    sig = auto_key(cache, sig, base)
    if sig is None:
        return None, None
    entry = cache.get(sig)
    if entry is None:            # first sweep of this structure: eager, watched by `backward`

        def watch(reason, independent=False):
            _remember(cache, sig, _Seen(independent) if reason is None else _Refused("backward sweep: " + reason))
        return None, watch
    if isinstance(entry, _Refused):
        return None, None
    if isinstance(entry, _Seen):
        try:
            if entry.independent and tuned_capture is not None:
                captured = tuned_capture()
            else:
                with _drift_then_diffusion(sde):
                    captured = capture()
        except Exception as e:
            captured = None
            cache[sig] = _Refused(f"capture failed: {type(e).__name__}: {e}")
        else:
            _remember(cache, sig, captured.accept() if captured is not None and captured.verified else
                      _Refused(_UNSTABLE if captured is not None and not captured.stable else
                               "the recorded sweep did not reproduce the eager one"))
        entry = cache[sig]
        return (entry if isinstance(entry, _CapturedBackward) else None), None
    return entry, None


# ---- back-propagation THROUGH the solver (sdeint with autograd) as two HIP graphs -------------------------------
class _CapturedTrainingSolve:
    """Forward solve recorded WITH its autograd graph, and the back-propagation through it, as two HIP graphs that
    share a memory pool (the scheme of ``torch.cuda.make_graphed_callables``, with the host-side planning kept
    outside and the Brownian seed in a device word).

    Static inputs: y0, the solver's initial extra state, the Brownian seed; the parameters are read in place (the
    module runs on leaf aliases of them while recording, see ``adjoint._capture_backward``). One forward replay
    must be followed by at most one backward replay before the next forward (the activations live in the pool)."""

    def __init__(self, solver, y0, ts, extra0, params):
        from torch.nn.utils.stateless import _reparametrize_module
        bm = solver.bm
        device = y0.device
        self.seed_dev = torch.zeros(1, dtype=torch.int64, device=device)
        self.y_in = torch.empty_like(y0, memory_format=torch.contiguous_format).requires_grad_(y0.requires_grad)
        self.extra_in = [torch.empty_like(e, memory_format=torch.contiguous_format).requires_grad_(e.requires_grad)
                         for e in extra0]
        self._load(bm, y0, extra0)
        alias_of = {id(p): p.detach().requires_grad_(True) for p in params}
        swapped = {name: alias_of[id(p)] for name, p in solver.sde.named_parameters(remove_duplicate=False)
                   if id(p) in alias_of}
        self.n_params = len(params)
        leaves = [x for x in [self.y_in] + self.extra_in if x.requires_grad] + [alias_of[id(p)] for p in params]
        self.leaf_is_input = [x.requires_grad for x in [self.y_in] + self.extra_in]
        self._keepalive = bm
        bm._entropy_dev = self.seed_dev

        def forward():
            solver._extra = tuple(self.extra_in)
            ys = solver._run(self.plan, self.y_in)
            return [ys] + list(solver._extra)

        def backward(outs, cotangents):
            live = [(o, c) for o, c in zip(outs, cotangents) if o.requires_grad]
            grads = torch.autograd.grad([o for o, _ in live], leaves, grad_outputs=[c for _, c in live],
                                        allow_unused=True)
            return [torch.zeros_like(x) if g is None else g for g, x in zip(grads, leaves)]

        try:
            with torch.enable_grad(), _reparametrize_module(solver.sde, swapped):
                self.plan = solver._plan(self.y_in, ts)
                side = torch.cuda.Stream(device=device)
                side.wait_stream(torch.cuda.current_stream(device))
                with torch.cuda.stream(side):       # warm-up outside capture (lazy inits, allocator)
                    outs = forward()
                    if _has_foreign_leaves(outs, leaves):
                        raise _NotCapturable("the solve depends on trainable tensors that are neither y0 nor "
                                             "parameters of the SDE module")
                    backward(outs, [torch.zeros_like(o) for o in outs])
                    del outs
                torch.cuda.current_stream(device).wait_stream(side)
                self.fwd_graph, self.bwd_graph = new_graph(), new_graph()
                with _capturing(self.fwd_graph, device):
                    self.outs = forward()
                self.cotangents = [torch.zeros_like(o) for o in self.outs]
                with _capturing(self.bwd_graph, device, pool=self.fwd_graph.pool()):
                    self.grads = backward(self.outs, self.cotangents)
            # forward + backward replayed a few times with all-ones cotangents must keep giving the same gradients
            for c in self.cotangents:
                c.fill_(1.0)

            def both():
                self.fwd_graph.replay()
                self.bwd_graph.replay()
            both()

            def eagerly():
                with torch.enable_grad():
                    backward(forward(), self.cotangents)
            self.stable = replays_are_stable(both, lambda: list(self.outs) + list(self.grads), eagerly)
        finally:
            bm._entropy_dev = None

    def _load(self, bm, y0, extra0):
        with torch.no_grad():
            self.y_in.copy_(y0)
            for dst, src in zip(self.extra_in, extra0):
                dst.copy_(src)
        key = bm._key
        self.seed_dev.fill_(key - (1 << 64) if key >= (1 << 63) else key)


class _NotCapturable(Exception):
    pass


def _has_foreign_leaves(outs, leaves):
    """True if the autograd graph of `outs` reaches a trainable leaf outside `leaves`: replaying a recorded backward
    pass would silently drop its gradient, so such solves are not recorded."""
    known = {id(x) for x in leaves}
    seen = set()
    stack = [o.grad_fn for o in outs if o.grad_fn is not None]
    while stack:
        fn = stack.pop()
        if fn in seen:
            continue
        seen.add(fn)
        var = getattr(fn, "variable", None)
        if var is not None and id(var) not in known:
            return True
        stack.extend(f for f, _ in fn.next_functions if f is not None)
    return False


class _GraphedSolve(torch.autograd.Function):
    @staticmethod
    def forward(ctx, captured, bm, n_extra, y0, *extras_and_params):
        captured._load(bm, y0, extras_and_params[:n_extra])
        captured.fwd_graph.replay()
        captured.generation = getattr(captured, "generation", 0) + 1
        ctx.captured, ctx.n_extra, ctx.generation = captured, n_extra, captured.generation
        return tuple(o.detach().clone() for o in captured.outs)

    @staticmethod
    def backward(ctx, *cotangents):
        captured = ctx.captured
        if captured.generation != ctx.generation:
            raise RuntimeError("torchsde_amd: with options={'hip_graph': True} the activations of a solve live in the "
                               "graph's memory pool and were overwritten by a later solve of the same structure; call "
                               "backward() before solving again, or drop the option for this call.")
        for dst, src in zip(captured.cotangents, cotangents):
            if src is None:
                dst.zero_()
            else:
                dst.copy_(src)
        captured.bwd_graph.replay()
        grads = [g.clone() for g in captured.grads]
        out, k = [], 0
        for is_leaf in captured.leaf_is_input:       # y0, then the extras
            out.append(grads[k] if is_leaf else None)
            k += 1 if is_leaf else 0
        return (None, None, None) + tuple(out) + tuple(grads[k:])


def replay_or_capture_training(solver, y0, ts, extra0, params):
    """`sdeint` with autograd through the solver, as a forward graph and a backward graph; returns (ys, extras) that
    carry a grad_fn. Falls back (returns None) when the Brownian motion cannot be re-seeded through a device word."""
    bm = solver.bm
    if not isinstance(bm, BrownianInterval) or bm._rootW is not None or bm._rootH is not None:
        warnings.warn("hip_graph=True needs a torchsde_amd.BrownianInterval without pinned W/H; running eagerly.")
        return None
    from . import timegrid
    ts_host = timegrid.ts_to_host(ts)
    params = [p for p in params if p.requires_grad]
    named = {id(p) for _, p in solver.sde.named_parameters(remove_duplicate=False)}
    if any(id(p) not in named for p in params):
        return None
    base = solver.sde
    while hasattr(base, "_base_sde"):
        base = base._base_sde
    cache = _cache_of(base)
    if not bm.frozen:
        bm.adopt_grid(timegrid.build(ts_host, solver.dt).t_f64())
    sig = ("training",) + _signature(solver, y0, ts_host) + (
        y0.requires_grad, tuple(e.requires_grad for e in extra0), tuple((p.data_ptr(), tuple(p.shape)) for p in params))
    captured = cache.get(sig)
    if captured is None:
        try:
            # (drift and diffusion stay in sequence here: the recorded autograd graph of the forward solve is replayed
            #  by a second graph, and its stream semantics are those of one stream)
            parallel_allowed = getattr(solver.sde, "overlap_f_g", False)
            if parallel_allowed:
                solver.sde.overlap_f_g = False
            try:
                captured = _CapturedTrainingSolve(solver, y0, ts, extra0, params)
            finally:
                if parallel_allowed:
                    solver.sde.overlap_f_g = True
            if not captured.stable:
                raise _NotCapturable(_UNSTABLE)
        except _NotCapturable as e:
            warnings.warn(f"hip_graph=True: {e}; running eagerly.")
            _remember(cache, sig, _Refused(str(e)))
            return None
        _remember(cache, sig, captured)
    if isinstance(captured, _Refused):
        return None
    outs = _GraphedSolve.apply(captured, bm, len(extra0), y0, *extra0, *params)
    return outs[0], tuple(outs[1:])
