"""Expression programs compiled instead of interpreted.

`tsde_trajectory_prog_diag` runs drift and diffusion of a recognised user module as postfix programs through an interpreter in
the kernel (csrc/trajectory.hip, `ProgModel`): a compare tree and register shuffles per instruction, ~3 000 cycles per
wave-step where the hand-written affine kernel needs ~400. This module turns the SAME instruction words into straight-line HIP
code -- a struct with the interpreter's interface, `f<SLOT>(x)`, `g<SLOT>(x)`, `gdg<SLOT>(x, g, v)` -- and instantiates the
interpreter's own kernel (`trajectory_prog_kernel<T, METHOD, W, Model>`: same loop, same schemes, same generator) with it:
one operation of the user's code = one statement, in the order the interpreter would execute it, with the same functions and
`-ffp-contract=off`, hence THE SAME BITS (checked on first use: the specialised launch must equal the interpreter's with
`torch.equal`, or it is never used).

The translation unit is compiled with the toolchain's `hipcc` (~4 s) in a background thread and cached by the hash of its source
under ``~/.cache/torchsde_amd/specialised/`` (``TSDE_SPECIALISE_CACHE``); until the library is there -- and whenever there is no
compiler -- the interpreter runs. ``TSDE_SPECIALISE=0`` switches the mechanism off, ``TSDE_SPECIALISE=sync`` compiles in the
calling thread (tests, benchmarks). The reference has no counterpart: its loop (base_solver.py:114-149) calls the user's Python
code at every step.
"""
import ctypes
import hashlib
import os
import shutil
import subprocess
import threading

import torch

from . import _native

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
MODE = os.environ.get("TSDE_SPECIALISE", "1").strip().lower()          # "0" | "1" (background) | "sync"

_OPS = {0: "load", 1: "add", 2: "sub", 3: "rsub", 4: "mul", 5: "div", 6: "rdiv", 16: "neg", 17: "exp", 18: "log", 19: "sin",
        20: "cos", 21: "tanh", 22: "sigmoid", 23: "softplus", 24: "sqrt", 25: "abs", 26: "relu", 27: "reciprocal",
        28: "square", 29: "cube", 30: "dup"}
_SRC_STACK, _SRC_CONST, _SRC_STATE, _SRC_TIME = 0, 1, 2, 3
_UNARY = {
    "neg": "vmap({0}, [](T v) {{ return -v; }})", "exp": "vmap({0}, [](T v) {{ return exp(v); }})",
    "log": "vmap({0}, [](T v) {{ return log(v); }})", "sin": "vmap({0}, [](T v) {{ return sin(v); }})",
    "cos": "vmap({0}, [](T v) {{ return cos(v); }})", "tanh": "vmap({0}, [](T v) {{ return tanh(v); }})",
    "sigmoid": "vmap({0}, [](T v) {{ return (T)1 / ((T)1 + exp(-v)); }})",
    "softplus": "vmap({0}, [](T v) {{ return v > (T)20 ? v : log1p(exp(v)); }})",
    "sqrt": "vmap({0}, [](T v) {{ return sqrt(v); }})", "abs": "vmap({0}, [](T v) {{ return fabs(v); }})",
    "relu": "vmap({0}, [](T v) {{ return v > (T)0 ? v : (T)0; }})",
    "reciprocal": "vmap({0}, [](T v) {{ return (T)1 / v; }})", "square": "({0} * {0})", "cube": "(({0} * {0}) * {0})",
}
_BINARY = {"add": "({a} + {b})", "sub": "({a} - {b})", "rsub": "({b} - {a})", "mul": "({a} * {b})", "div": "({a} / {b})",
           "rdiv": "({b} / {a})"}


# the same functions on forward-mode dual numbers: value and slope exactly as ProgSensModel::run forms them
_DUAL_HELPERS = '''
template <typename T> TSDE_D Dual<T> d_neg(const Dual<T>& s) { return chain(s, -s.v, (T)-1); }
template <typename T> TSDE_D Dual<T> d_exp(const Dual<T>& s) { const T e = exp(s.v); return chain(s, e, e); }
template <typename T> TSDE_D Dual<T> d_log(const Dual<T>& s) { return chain(s, log(s.v), (T)1 / s.v); }
template <typename T> TSDE_D Dual<T> d_sin(const Dual<T>& s) { return chain(s, sin(s.v), cos(s.v)); }
template <typename T> TSDE_D Dual<T> d_cos(const Dual<T>& s) { return chain(s, cos(s.v), -sin(s.v)); }
template <typename T> TSDE_D Dual<T> d_tanh(const Dual<T>& s) { const T t = tanh(s.v); return chain(s, t, (T)1 - t * t); }
template <typename T> TSDE_D Dual<T> d_sigmoid(const Dual<T>& s) {
  const T g = (T)1 / ((T)1 + exp(-s.v));
  return chain(s, g, g * ((T)1 - g));
}
template <typename T> TSDE_D Dual<T> d_softplus(const Dual<T>& s) {
  const T v = s.v;
  return chain(s, v > (T)20 ? v : log1p(exp(v)), v > (T)20 ? (T)1 : (T)1 / ((T)1 + exp(-v)));
}
template <typename T> TSDE_D Dual<T> d_sqrt(const Dual<T>& s) { const T r = sqrt(s.v); return chain(s, r, (T)0.5 / r); }
template <typename T> TSDE_D Dual<T> d_abs(const Dual<T>& s) {
  const T v = s.v;
  return chain(s, fabs(v), v > (T)0 ? (T)1 : (v < (T)0 ? (T)-1 : (T)0));
}
template <typename T> TSDE_D Dual<T> d_relu(const Dual<T>& s) {
  const T v = s.v;
  return chain(s, v > (T)0 ? v : (T)0, v > (T)0 ? (T)1 : (T)0);
}
template <typename T> TSDE_D Dual<T> d_reciprocal(const Dual<T>& s) { const T r = (T)1 / s.v; return chain(s, r, -(r * r)); }
template <typename T> TSDE_D Dual<T> d_square(const Dual<T>& s) { const T v = s.v; return chain(s, v * v, (T)2 * v); }
template <typename T> TSDE_D Dual<T> d_cube(const Dual<T>& s) { const T v = s.v; return chain(s, (v * v) * v, (T)3 * (v * v)); }
'''


def _body(words, name, dual=False):
    """Straight-line code for one program: the interpreter's stack machine (csrc/trajectory.hip ProgModel::run; `dual`:
    ProgSensModel::run) run at generation time, every instruction one `const V` statement. Returns (lines, result expression,
    constants used)."""
    lines, stack, used = [], [], set()
    count = 0
    vtype = "S" if dual else "V"

    def fresh(expr):
        nonlocal count
        var = f"{name}{count}"
        count += 1
        lines.append(f"    const {vtype} {var} = {expr};")
        return var
    for ins in words:
        op, src, k = _OPS[ins & 0xFF], (ins >> 8) & 0xFF, ins >> 16
        if (ins & 0xFF) < 16:
            if src == _SRC_STACK:
                b = stack.pop()
                a = stack.pop()
                stack.append(fresh(_BINARY[op].format(a=a, b=b)))
                continue
            if src == _SRC_CONST:
                used.add(k)
                operand = f"c{k}"
            elif src == _SRC_TIME:
                operand = f"{vtype}(time)"
            else:
                operand = "x"
            if op == "load":
                stack.append(operand)
            else:
                a = stack.pop()
                stack.append(fresh(_BINARY[op].format(a=a, b=operand)))
        elif op == "dup":
            stack.append(stack[-1])
        elif dual:
            stack.append(fresh(f"d_{op}({stack.pop()})"))
        else:
            stack.append(fresh(_UNARY[op].format(stack.pop())))
    if not stack:
        return lines, f"{vtype}((T)0)", used
    return lines, stack[-1], used


def source(f_code, g_code, dg_code, n_const, dtype, method, kind="values"):
    """The translation unit for these programs, this state dtype and this scheme. `kind`: "values"
    (`trajectory_prog_kernel`), or "sens" (`trajectory_prog_sens_kernel`: the programs on dual numbers)."""
    if kind == "sens":
        return _source_sens(f_code, g_code, dg_code, n_const, dtype, method)
    additive = kind.startswith("additive")
    ctype = "float" if dtype == torch.float32 else "double"
    parts, used = {}, set()
    for name, words in (("f", f_code), ("g", g_code), ("h", dg_code)):
        lines, result, consts = _body(tuple(words), name)
        parts[name] = (lines, result)
        used |= consts
    used = sorted(used)
    members = "".join(f"  V c{k};\n" for k in used)
    setup = "".join(
        f"    {{ const Pack<T, W> pk = load<T, W>(p.consts, (int64_t){k} * p.d + column);\n"
        f"      _Pragma(\"unroll\") for (int q = 0; q < W; ++q) c{k}.v[q] = pk.v[q]; }}\n" for k in used)

    def fn(name):
        lines, result = parts[name]
        return "\n".join(lines) + ("\n" if lines else "") + f"    return {result};"
    return f'''// generated by torchsde_amd/specialise.py -- do not edit
#define TSDE_SPECIALISE_TU 1
#include "{os.path.join(_CSRC, "trajectory.hip")}"
namespace tsde {{
template <typename T, int W>
struct SpecModel {{
  using V = Vec<T, W>;
{members}  T tslot[4];
  TSDE_D void setup(const ProgArgs<T>& p, int64_t column) {{
{setup}  }}
  TSDE_D V eval_f(const V& x, const T time) const {{
{fn("f")}
  }}
  TSDE_D V eval_g(const V& x, const T time) const {{
{fn("g")}
  }}
  TSDE_D V eval_h(const V& x, const T time) const {{
{fn("h")}
  }}
  template <int SLOT>
  TSDE_D V f(const V& x) const {{ return eval_f(x, tslot[SLOT]); }}
  template <int SLOT>
  TSDE_D V g(const V& x) const {{ return eval_g(x, tslot[SLOT]); }}
  template <int SLOT>
  TSDE_D V gdg(const V& x, const V& gv, const V& v2) const {{ return (gv * v2) * eval_h(x, tslot[SLOT]); }}
}};
}}  // namespace tsde

{_additive_launcher(ctype, method, kind, used) if additive else ""}
extern "C" int tsde_specialised_launch(void* ys, const void* y0, int64_t rows, int64_t d, const void* consts, int n_const,
                                       int scalar_noise, const tsde_traj_t* tr, uint64_t entropy, uint64_t elem0,
                                       const uint64_t* entropy_dev, void* stream) {{
  using namespace tsde;
  if ({1 if additive else 0}) return (int)hipErrorInvalidValue;        // (an additive-noise unit: tsde_specialised_additive_launch)
  using T = {ctype};
  constexpr int METHOD = {int(method)};
  if (n_const < {(used[-1] + 1) if used else 0}) return (int)hipErrorInvalidValue;
  ProgArgs<T> p;
  p.ys = (T*)ys;
  p.y0 = (const T*)y0;
  p.consts = (const T*)consts;
  p.f_len = p.g_len = p.dg_len = 0;
  p.n_const = n_const;
  p.scalar_noise = scalar_noise;
  p.rows = (const T*)tr->step_rows;
  p.cells = tr->cells;
  p.out_step = tr->out_step;
  p.out_w = (const T*)tr->out_w;
  p.n = rows * d;
  p.d = d;
  p.n_steps = tr->n_steps;
  p.n_out = tr->n_out;
  p.key.k0 = (uint32_t)entropy;
  p.key.k1 = (uint32_t)(entropy >> 32);
  p.key.elem0 = elem0;
  p.key_dev = entropy_dev;
  for (int w = 0; w < kProgWords; ++w) p.code[w] = 0u;
  if (p.n <= 0 || p.n_steps <= 0) return 0;
  const hipStream_t s = (hipStream_t)stream;
  const bool can_vec = (d % 4 == 0) && (scalar_noise || elem0 % 4 == 0) && aligned16(ys) && aligned16(y0) &&
                       ((p.n * sizeof(T)) % 16 == 0);
  const bool vec = can_vec && (p.n >> 2) >= kTrajVecMinGroups;
  if (vec) {{
    const int64_t lanes = p.n >> 2;
    hipLaunchKernelGGL((trajectory_prog_kernel<T, METHOD, 4, SpecModel<T, 4>>), dim3((unsigned)((lanes + kBlock - 1) / kBlock)),
                       dim3(kBlock), 0, s, p);
  }} else {{
    hipLaunchKernelGGL((trajectory_prog_kernel<T, METHOD, 1, SpecModel<T, 1>>), dim3((unsigned)((p.n + kBlock - 1) / kBlock)),
                       dim3(kBlock), 0, s, p);
  }}
  return (int)hipGetLastError();
}}
'''


def _additive_launcher(ctype, method, kind, used):
    """The launcher of an additive-noise unit (kind "additive<MP>"): csrc/trajectory.hip launch_trajectory_prog_additive for ONE
    scheme and ONE channel-count class, with the generated drift model."""
    mp = int(kind[len("additive"):])
    return f'''
extern "C" int tsde_specialised_additive_launch(void* ys, const void* y0, int64_t rows, int64_t d, int64_t m, const void* consts,
                                                int n_const, const void* gtab, int time_dependent, const tsde_traj_t* tr,
                                                uint64_t entropy, uint64_t elem0, const uint64_t* entropy_dev, void* stream) {{
  using namespace tsde;
  using T = {ctype};
  constexpr int METHOD = {int(method)};
  constexpr int MP = {mp};
  if (n_const < {(used[-1] + 1) if used else 0} || m < 1 || m > MP) return (int)hipErrorInvalidValue;
  ProgAdditiveArgs<T> q;
  ProgArgs<T>& p = q.base;
  p.ys = (T*)ys;
  p.y0 = (const T*)y0;
  for (int w = 0; w < kProgWords; ++w) p.code[w] = 0u;
  p.consts = (const T*)consts;
  p.f_len = p.g_len = p.dg_len = 0;
  p.n_const = n_const;
  p.scalar_noise = 0;
  p.rows = (const T*)tr->step_rows;
  p.cells = tr->cells;
  p.out_step = tr->out_step;
  p.out_w = (const T*)tr->out_w;
  p.n = rows * d;
  p.d = d;
  p.n_steps = tr->n_steps;
  p.n_out = tr->n_out;
  p.key.k0 = (uint32_t)entropy;
  p.key.k1 = (uint32_t)(entropy >> 32);
  p.key.elem0 = elem0;
  p.key_dev = entropy_dev;
  if (p.n <= 0 || p.n_steps <= 0) return 0;
  const int slots = METHOD == kEuler ? 1 : 2;
  q.gtab = (const T*)gtab;
  q.slot_stride = time_dependent ? m * d : 0;
  q.step_stride = time_dependent ? (int64_t)slots * m * d : 0;
  q.m = (int32_t)m;
  q.quads = (m % 4 == 0 && elem0 % 4 == 0) ? 1 : 0;
  const bool can_vec = (d % 4 == 0) && aligned16(ys) && aligned16(y0) && aligned16(gtab) && ((p.n * sizeof(T)) % 16 == 0);
  const bool vec = can_vec && (p.n >> 2) >= kTrajVecMinGroups;
  const hipStream_t s = (hipStream_t)stream;
  if (vec) {{
    hipLaunchKernelGGL((trajectory_prog_additive_kernel<T, METHOD, 4, MP, SpecModel<T, 4>>),
                       dim3((unsigned)(((p.n >> 2) + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, q);
  }} else {{
    hipLaunchKernelGGL((trajectory_prog_additive_kernel<T, METHOD, 1, MP, SpecModel<T, 1>>),
                       dim3((unsigned)((p.n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, q);
  }}
  return (int)hipGetLastError();
}}
'''


def _source_sens(f_code, g_code, dg_code, n_const, dtype, method):
    ctype = "float" if dtype == torch.float32 else "double"
    parts, used = {}, set()
    for name, words in (("f", f_code), ("g", g_code), ("h", dg_code)):
        lines, result, consts = _body(tuple(words), name, dual=True)
        parts[name] = (lines, result)
        used |= consts
    used = sorted(used)
    members = "".join(f"  S c{k};\n" for k in used)
    setup = "".join(
        f"    c{k} = S(q.base.consts[(int64_t){k} * q.base.d + column]);\n"
        f"    if (q.param_slot[{k}] > 0) c{k}.d[q.param_slot[{k}]] = (T)1;\n" for k in used)

    def fn(name):
        lines, result = parts[name]
        return "\n".join(lines) + ("\n" if lines else "") + f"    return {result};"
    return f'''// generated by torchsde_amd/specialise.py -- do not edit
#define TSDE_SPECIALISE_TU 1
#include "{os.path.join(_CSRC, "trajectory.hip")}"
namespace tsde {{
{_DUAL_HELPERS}
template <typename T>
struct SpecSensModel {{
  using S = Dual<T>;
{members}  T tslot[4];
  TSDE_D void setup(const ProgSensArgs<T>& q, int64_t column) {{
{setup}  }}
  TSDE_D S eval_f(const S& x, const T time) const {{
{fn("f")}
  }}
  TSDE_D S eval_g(const S& x, const T time) const {{
{fn("g")}
  }}
  TSDE_D S eval_h(const S& x, const T time) const {{
{fn("h")}
  }}
  template <int SLOT>
  TSDE_D S f(const S& x) const {{ return eval_f(x, tslot[SLOT]); }}
  template <int SLOT>
  TSDE_D S g(const S& x) const {{ return eval_g(x, tslot[SLOT]); }}
  template <int SLOT>
  TSDE_D S gdg(const S& x, const S& gv, T v2) const {{ return (gv * v2) * eval_h(x, tslot[SLOT]); }}
}};
}}  // namespace tsde

extern "C" int tsde_specialised_sens_launch(void* ys, void* sens, const int8_t* param_slot, const void* y0, int64_t rows,
                                            int64_t d, const void* consts, int n_const, int scalar_noise, const tsde_traj_t* tr,
                                            uint64_t entropy, uint64_t elem0, const uint64_t* entropy_dev, void* stream) {{
  using namespace tsde;
  using T = {ctype};
  constexpr int METHOD = {int(method)};
  if (n_const < {(used[-1] + 1) if used else 0} || n_const > kProgParamRows) return (int)hipErrorInvalidValue;
  ProgSensArgs<T> q;
  ProgArgs<T>& p = q.base;
  p.ys = (T*)ys;
  p.y0 = (const T*)y0;
  p.consts = (const T*)consts;
  p.f_len = p.g_len = p.dg_len = 0;
  p.n_const = n_const;
  p.scalar_noise = scalar_noise;
  p.rows = (const T*)tr->step_rows;
  p.cells = tr->cells;
  p.out_step = tr->out_step;
  p.out_w = (const T*)tr->out_w;
  p.n = rows * d;
  p.d = d;
  p.n_steps = tr->n_steps;
  p.n_out = tr->n_out;
  p.key.k0 = (uint32_t)entropy;
  p.key.k1 = (uint32_t)(entropy >> 32);
  p.key.elem0 = elem0;
  p.key_dev = entropy_dev;
  for (int w = 0; w < kProgWords; ++w) p.code[w] = 0u;
  q.sens = (T*)sens;
  for (int k = 0; k < kProgParamRows; ++k) q.param_slot[k] = (param_slot && k < n_const) ? param_slot[k] : (int8_t)-1;
  if (p.n <= 0 || p.n_steps <= 0) return 0;
  hipLaunchKernelGGL((trajectory_prog_sens_kernel<T, METHOD, SpecSensModel<T>>), dim3((unsigned)((p.n + kBlock - 1) / kBlock)),
                     dim3(kBlock), 0, (hipStream_t)stream, q);
  return (int)hipGetLastError();
}}
'''


# ---- compiling, caching, loading -------------------------------------------------------------------------------------------
_lock = threading.Lock()
_state = {}            # key -> "pending" | "failed: ..." | _Library
_verified = {}         # key -> True | False (the specialised launch reproduced the interpreter bit for bit)


class _Library:
    def __init__(self, path):
        self.path = path
        self.lib = ctypes.CDLL(path)
        tail = [ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_native.Traj),
                ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p]
        if hasattr(self.lib, "tsde_specialised_rows_launch"):
            fn = self.lib.tsde_specialised_rows_launch
            fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int,
                           ctypes.POINTER(_native.Traj), ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p]
        elif hasattr(self.lib, "tsde_specialised_additive_launch"):
            fn = self.lib.tsde_specialised_additive_launch
            fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p,
                           ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(_native.Traj), ctypes.c_uint64,
                           ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p]
        elif hasattr(self.lib, "tsde_specialised_sens_launch"):
            fn = self.lib.tsde_specialised_sens_launch
            fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p] + tail
        else:
            fn = self.lib.tsde_specialised_launch
            fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + tail
        fn.restype = ctypes.c_int
        self.launch = fn


def cache_dir():
    """Where the compiled units live: ``TSDE_SPECIALISE_CACHE``, else under ``~/.cache``; a home that cannot be written to
    (a container's read-only user) falls back to the system's temporary directory."""
    import tempfile
    wanted = os.environ.get("TSDE_SPECIALISE_CACHE") or os.path.join(os.path.expanduser("~"), ".cache", "torchsde_amd", "specialised")
    for root in (wanted, os.path.join(tempfile.gettempdir(), f"torchsde_amd_specialised_{os.getuid()}")):
        try:
            os.makedirs(root, exist_ok=True)
            if os.access(root, os.W_OK):
                return root
        except OSError:
            continue
    raise OSError(f"no writable directory for compiled programs (tried {wanted} and the temporary directory)")


def compiler():
    return os.environ.get("HIPCC") or shutil.which("hipcc") or ("/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc")
                                                               else None)


def _arch(device):
    try:
        return torch.cuda.get_device_properties(device).gcnArchName.split(":")[0]
    except Exception:
        return "gfx950"


def _compile(key, text, arch):
    try:
        path = os.path.join(cache_dir(), f"{key}.so")
        if not os.path.exists(path):
            src = os.path.join(cache_dir(), f"{key}.hip")
            with open(src, "w") as fh:
                fh.write(text)
            tmp = f"{path}.{os.getpid()}.{threading.get_ident()}.tmp"
            cmd = [compiler(), "-O3", "-std=c++17", "-fPIC", f"--offload-arch={arch}", "-ffp-contract=off", "-shared", "-o", tmp, src]
            done = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
            if done.returncode != 0:
                raise RuntimeError(done.stderr[-2000:])
            os.replace(tmp, path)                 # (atomic: another process may be compiling the same key)
        result = _Library(path)
    except Exception as e:       # no compiler, a compile error, a load error: the interpreter stays
        result = f"failed: {type(e).__name__}: {e}"
    with _lock:
        _state[key] = result


def lookup(f_code, g_code, dg_code, n_const, dtype, method, device, wait=None, kind="values"):
    """(key, the loaded library of these programs or None). The first call starts the compilation (in the background unless
    TSDE_SPECIALISE=sync or `wait`)."""
    if MODE in ("0", "false", "off") or compiler() is None:
        return None, None
    arch = _arch(device)
    text = source(f_code, g_code, dg_code, n_const, dtype, method, kind)
    key = hashlib.sha256((text + arch + _sources_digest()).encode()).hexdigest()[:24]
    with _lock:
        have = _state.get(key)
        if have is None:
            _state[key] = "pending"
    if have is None:
        if MODE == "sync" or wait:
            _compile(key, text, arch)
        elif not _enqueue(key, text, arch):
            with _lock:
                _state.pop(key, None)            # (the queue is full: ask again at a later solve)
        with _lock:
            have = _state.get(key)
    return key, (have if isinstance(have, _Library) else None)


# one worker, a short queue: a process that meets hundreds of different programs (a test-suite) compiles a few at a time
_queue = None


def _enqueue(key, text, arch):
    global _queue
    import queue
    with _lock:
        if _queue is None:
            _queue = queue.Queue(maxsize=8)

            def work():
                while True:
                    job = _queue.get()
                    _compile(*job)
            threading.Thread(target=work, daemon=True, name="torchsde_amd-specialise").start()
    try:
        _queue.put_nowait((key, text, arch))
        return True
    except queue.Full:
        return False


_digest = None


def _sources_digest():
    """Hash of the kernel sources a specialised unit includes: a changed header must not find a stale cached library."""
    global _digest
    if _digest is None:
        h = hashlib.sha256()
        for name in sorted(os.listdir(_CSRC)):
            if name.endswith((".h", ".hip")):
                with open(os.path.join(_CSRC, name), "rb") as fh:
                    h.update(fh.read())
        with open(os.path.join(os.path.dirname(_HERE), "include", "torchsde_amd.h"), "rb") as fh:
            h.update(fh.read())
        _digest = h.hexdigest()
    return _digest


def status():
    """{key: "pending" | "failed: ..." | path} of every program seen by this process (diagnostics)."""
    with _lock:
        return {k: (v.path if isinstance(v, _Library) else v) for k, v in _state.items()}


def launch(library, ys, y0, consts, scalar_noise, schedule, bm, stream):
    rows, d = y0.shape
    entropy_dev = bm._entropy_dev
    lib = _native.load()
    slot = lib.tsde_prof_bracket_open(_native.KID_TRAJECTORY, stream)       # (bench.py's per-launch timing of this kernel family)
    rc = library.launch(ys.data_ptr(), y0.data_ptr(), rows, d, consts.data_ptr(), consts.shape[0], int(bool(scalar_noise)),
                        schedule.struct(), bm._key, bm._elem0, None if entropy_dev is None else entropy_dev.data_ptr(), stream)
    if slot >= 0:
        lib.tsde_prof_bracket_close(slot, stream)
    if rc != 0:
        raise _native.NativeLibraryError(f"torchsde_amd: a specialised program kernel failed with hipError {rc}")


def verified(key):
    return _verified.get(key)


def set_verified(key, ok):
    _verified[key] = bool(ok)


def launch_sens(library, ys, sens, slots, y0, consts, n_const, scalar_noise, schedule, bm, stream):
    rows, d = y0.shape
    entropy_dev = bm._entropy_dev
    lib = _native.load()
    slot = lib.tsde_prof_bracket_open(_native.KID_TRAJECTORY, stream)
    rc = library.launch(ys.data_ptr(), sens.data_ptr(), ctypes.cast(slots, ctypes.c_void_p), y0.data_ptr(), rows, d,
                        consts.data_ptr(), int(n_const), int(bool(scalar_noise)), schedule.struct(), bm._key, bm._elem0,
                        None if entropy_dev is None else entropy_dev.data_ptr(), stream)
    if slot >= 0:
        lib.tsde_prof_bracket_close(slot, stream)
    if rc != 0:
        raise _native.NativeLibraryError(f"torchsde_amd: a specialised program kernel failed with hipError {rc}")


def launch_additive(library, ys, y0, consts, g_table, m, timed, schedule, bm, stream):
    rows, d = y0.shape
    entropy_dev = bm._entropy_dev
    lib = _native.load()
    slot = lib.tsde_prof_bracket_open(_native.KID_TRAJECTORY, stream)
    rc = library.launch(ys.data_ptr(), y0.data_ptr(), rows, d, int(m), consts.data_ptr(), consts.shape[0], g_table.data_ptr(),
                        int(bool(timed)), schedule.struct(), bm._key, bm._elem0,
                        None if entropy_dev is None else entropy_dev.data_ptr(), stream)
    if slot >= 0:
        lib.tsde_prof_bracket_close(slot, stream)
    if rc != 0:
        raise _native.NativeLibraryError(f"torchsde_amd: a specialised program kernel failed with hipError {rc}")


# ---- row-coupled systems: a lane owns a whole row (recognise_rows.py) --------------------------------------------------------
_ROW_OPS = {
    "add": "({0} + {1})", "sub": "({0} - {1})", "mul": "({0} * {1})", "div": "({0} / {1})", "neg": "(-{0})",
    "exp": "exp({0})", "log": "log({0})", "sin": "sin({0})", "cos": "cos({0})", "tanh": "tanh({0})",
    "sigmoid": "((T)1 / ((T)1 + exp(-{0})))", "softplus": "({0} > (T)20 ? {0} : log1p(exp({0})))", "sqrt": "sqrt({0})",
    "abs": "fabs({0})", "relu": "({0} > (T)0 ? {0} : (T)0)", "reciprocal": "((T)1 / {0})", "square": "({0} * {0})",
    "cube": "(({0} * {0}) * {0})",
}


def source_rows(structure, n_const, dtype, method):
    """The translation unit of a row-coupled system: `structure` = RecognisedRows.structure()."""
    (_, d, statements, outputs), _ = structure
    ctype = "float" if dtype == torch.float32 else "double"
    names = [f"n{k}" for k in range(len(statements))]
    needs = {}
    for name, (op, operands) in zip(names, statements):
        needs[name] = set(o for o in operands if o in needs or o.startswith("n"))

    def body(outs):
        wanted, stack = set(), [o for o in outs if o.startswith("n")]
        while stack:
            x = stack.pop()
            if x in wanted:
                continue
            wanted.add(x)
            stack.extend(o for o in needs.get(x, ()) if o.startswith("n"))
        lines = [f"    const T {name} = {_ROW_OPS[op].format(*operands)};" for name, (op, operands) in zip(names, statements)
                 if name in wanted]
        lines.append("    V r;")
        lines += [f"    r.v[{c}] = {o};" for c, o in enumerate(outs)]
        lines.append("    return r;")
        return "\n".join(lines)
    nc = max(1, n_const)
    return f'''// generated by torchsde_amd/specialise.py (a row-coupled system) -- do not edit
#define TSDE_SPECIALISE_TU 1
#include "{os.path.join(_CSRC, "trajectory.hip")}"
namespace tsde {{
template <typename T>
struct RowModel {{
  using V = Vec<T, {d}>;
  T c[{nc}];
  T tslot[4];
  TSDE_D void setup(const ProgArgs<T>& p, int64_t) {{
    _Pragma("unroll") for (int k = 0; k < {nc}; ++k) c[k] = k < p.n_const ? p.consts[k] : (T)0;
  }}
  TSDE_D V eval_f(const V& x, const T time) const {{
{body(list(outputs[:d]))}
  }}
  TSDE_D V eval_g(const V& x, const T time) const {{
{body(list(outputs[d:]))}
  }}
  template <int SLOT>
  TSDE_D V f(const V& x) const {{ return eval_f(x, tslot[SLOT]); }}
  template <int SLOT>
  TSDE_D V g(const V& x) const {{ return eval_g(x, tslot[SLOT]); }}
  template <int SLOT>
  TSDE_D V gdg(const V& x, const V& gv, const V& v2) const {{ return V((T)0); }}        // (no derivative schemes on this route)
}};
}}  // namespace tsde

extern "C" int tsde_specialised_rows_launch(void* ys, const void* y0, int64_t rows, int64_t d, const void* consts, int n_const,
                                            const tsde_traj_t* tr, uint64_t entropy, uint64_t elem0,
                                            const uint64_t* entropy_dev, void* stream) {{
  using namespace tsde;
  using T = {ctype};
  constexpr int METHOD = {int(method)};
  if (d != {d} || n_const > {nc}) return (int)hipErrorInvalidValue;
  ProgArgs<T> p;
  p.ys = (T*)ys;
  p.y0 = (const T*)y0;
  p.consts = (const T*)consts;
  p.f_len = p.g_len = p.dg_len = 0;
  p.n_const = n_const;
  p.scalar_noise = 0;
  p.rows = (const T*)tr->step_rows;
  p.cells = tr->cells;
  p.out_step = tr->out_step;
  p.out_w = (const T*)tr->out_w;
  p.n = rows * d;
  p.d = d;
  p.n_steps = tr->n_steps;
  p.n_out = tr->n_out;
  p.key.k0 = (uint32_t)entropy;
  p.key.k1 = (uint32_t)(entropy >> 32);
  p.key.elem0 = elem0;
  p.key_dev = entropy_dev;
  for (int w = 0; w < kProgWords; ++w) p.code[w] = 0u;
  if (p.n <= 0 || p.n_steps <= 0) return 0;
  hipLaunchKernelGGL((trajectory_prog_kernel<T, METHOD, {d}, RowModel<T>>), dim3((unsigned)((rows + kBlock - 1) / kBlock)),
                     dim3(kBlock), 0, (hipStream_t)stream, p);
  return (int)hipGetLastError();
}}
'''


def lookup_rows(structure, n_const, dtype, method, device, wait=None):
    """(key, library or None) of a row-coupled system; starts the compilation on first sight (cf. `lookup`)."""
    if MODE in ("0", "false", "off") or compiler() is None:
        return None, None
    arch = _arch(device)
    text = source_rows(structure, n_const, dtype, method)
    key = hashlib.sha256((text + arch + _sources_digest()).encode()).hexdigest()[:24]
    with _lock:
        have = _state.get(key)
        if have is None:
            _state[key] = "pending"
    if have is None:
        if MODE == "sync" or wait:
            _compile(key, text, arch)
        elif not _enqueue(key, text, arch):
            with _lock:
                _state.pop(key, None)
        with _lock:
            have = _state.get(key)
    return key, (have if isinstance(have, _Library) else None)


def launch_rows(library, ys, y0, consts, schedule, bm, stream):
    rows, d = y0.shape
    entropy_dev = bm._entropy_dev
    lib = _native.load()
    slot = lib.tsde_prof_bracket_open(_native.KID_TRAJECTORY, stream)
    rc = library.launch(ys.data_ptr(), y0.data_ptr(), rows, d, consts.data_ptr(), consts.numel(), schedule.struct(), bm._key,
                        bm._elem0, None if entropy_dev is None else entropy_dev.data_ptr(), stream)
    if slot >= 0:
        lib.tsde_prof_bracket_close(slot, stream)
    if rc != 0:
        raise _native.NativeLibraryError(f"torchsde_amd: a specialised row kernel failed with hipError {rc}")
