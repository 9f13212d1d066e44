"""The reversible Heun pair for neural SDEs on the matrix cores: ``sdeint(..., method="reversible_heun")`` and
``sdeint_adjoint(..., method="reversible_heun", adjoint_method="adjoint_reversible_heun")`` -- what the reference "strongly
recommends" for adjoint training (DOCUMENTATION.md:97,118) and the only pair examples/sde_gan.py:129-130 uses -- for modules whose
drift and diffusion are perceptrons of (t, y) of up to four Linear layers (csrc/tsde_neural_rheun.h).

Forward: ONE launch of ``tsde_rheun_mlp_forward`` (methods/reversible_heun.py:48-73 driven by base_solver.py:114-134). Backward:
``tsde_rheun_mlp_backward`` over chunks of evaluations (methods/reversible_heun.py:76-144 driven by adjoint.py:64-127: the forward
state is reconstructed algebraically, the vector-Jacobian products of adjoint_sde.py run on the matrix cores), each followed by
the weight-gradient products over that chunk's stash -- tall matrix products, which go to the BLAS library. Memory: the outputs
plus O(chunk) stash, independent of the number of steps.

`DeepNet` is the kernels' view of one perceptron (``tsde_deep_mlp_t``), built from the tensors an ``nn.Sequential`` of
``nn.Linear`` layers holds; `ReversibleHeunFn` is the autograd function both routes go through.
"""
import ctypes

import numpy as np
import torch

from . import _native
from . import kernels as K

MAX_MID = 2
STASH_BYTES = 2 << 30            # per chunk of the backward sweep


class DeepNet:
    """out = scale * final(lin_last(a(... a(lin_first(cat([t, y]) or y)) ...))),  a = act_scale * act.

    `linears`: [(weight (out, in), bias or None), ...] exactly as ``nn.Linear`` holds them, 2 to 4 of them; the first layer's
    input is the state, or -- `time_input` -- ``cat([t.expand(B, 1), y], 1)`` (column 0 of its weight multiplies t)."""

    def __init__(self, linears, activation, act_scale=1.0, final=_native.FINAL_NONE, scale=1.0, time_input=False):
        if not 2 <= len(linears) <= 2 + MAX_MID:
            raise ValueError("a perceptron of 2 to 4 Linear layers")
        self.linears = [(w, b) for w, b in linears]
        self.activation, self.act_scale, self.final, self.scale = int(activation), float(act_scale), int(final), float(scale)
        self.time_input = bool(time_input)
        self.hidden = int(linears[0][0].shape[0])
        self.out = int(linears[-1][0].shape[0])
        self.n_mid = len(linears) - 2
        self.d = int(linears[0][0].shape[1]) - (1 if time_input else 0)
        for w, _ in linears[1:-1]:
            if tuple(w.shape) != (self.hidden, self.hidden):
                raise ValueError("hidden layers of one width")
        if linears[-1][0].shape[1] != self.hidden:
            raise ValueError("the last layer must act on the hidden units")
        self._device_tensors = None

    def parameters(self):
        """The user's tensors in a fixed order: weight, bias (skipped when None) of every layer."""
        return [t for w, b in self.linears for t in (w, b) if t is not None]

    def structure(self):
        return (tuple(tuple(w.shape) + (b is not None,) for w, b in self.linears), self.activation, self.act_scale, self.final,
                self.scale, self.time_input)

    def device_tensors(self):
        """Contiguous float32 copies in the kernels' input-major layout (kept alive for the launch)."""
        if self._device_tensors is None:
            def bias(b, n, like):
                return torch.zeros(n, dtype=torch.float32, device=like.device) if b is None else b.detach().contiguous()
            w1, b1 = self.linears[0]
            w1 = w1.detach()
            t = dict(w1t=w1[:, 0].contiguous() if self.time_input else None,
                     w1=(w1[:, 1:] if self.time_input else w1).t().contiguous(), b1=bias(b1, self.hidden, w1),
                     wm=[w.detach().t().contiguous() for w, _ in self.linears[1:-1]],
                     bm=[bias(b, self.hidden, w) for w, b in self.linears[1:-1]],
                     w2=self.linears[-1][0].detach().t().contiguous(), b2=bias(self.linears[-1][1], self.out, w1))
            self._device_tensors = t
        return self._device_tensors

    def struct(self):
        t = self.device_tensors()
        s = _native.DeepMlp()
        s.w1, s.b1, s.w2, s.b2 = t["w1"].data_ptr(), t["b1"].data_ptr(), t["w2"].data_ptr(), t["b2"].data_ptr()
        s.w1t = 0 if t["w1t"] is None else t["w1t"].data_ptr()
        for l in range(MAX_MID):
            s.wm[l] = t["wm"][l].data_ptr() if l < self.n_mid else 0
            s.bm[l] = t["bm"][l].data_ptr() if l < self.n_mid else 0
        s.hidden, s.out, s.activation, s.final, s.n_mid, s.reserved = self.hidden, self.out, self.activation, self.final, self.n_mid, 0
        s.scale, s.act_scale = self.scale, self.act_scale
        return s

    def rebuilt(self, tensors):
        """The same net over `tensors` (in `parameters()` order): the autograd function's own view of its inputs."""
        it = iter(tensors)
        linears = [(next(it), None if b is None else next(it)) for _, b in self.linears]
        return DeepNet(linears, self.activation, self.act_scale, self.final, self.scale, self.time_input)

    # ---- a torch statement of the same function (tests, and the last layer of a general diffusion on the way back) ----------
    def hidden_act(self, x):
        if self.activation == _native.ACT_TANH:
            v = torch.tanh(x)
        elif self.activation == _native.ACT_SOFTPLUS:
            v = torch.nn.functional.softplus(x)
        else:
            v = torch.nn.functional.silu(x)
        return v if self.act_scale == 1.0 else self.act_scale * v

    def final_act(self, x):
        if self.final == _native.FINAL_SIGMOID:
            return torch.sigmoid(x)
        if self.final == _native.FINAL_TANH:
            return torch.tanh(x)
        return x

    def __call__(self, t, y):
        x = torch.cat([t.to(y.dtype).expand(y.shape[0], 1), y], dim=1) if self.time_input else y
        for w, b in self.linears[:-1]:
            x = self.hidden_act(torch.nn.functional.linear(x, w, b))
        w, b = self.linears[-1]
        return self.scale * self.final_act(torch.nn.functional.linear(x, w, b))


def lds_bytes(d, m, drift, diffusion, noise):
    """LDS bytes the kernels need for this pair of nets (0: no kernel covers the shape)."""
    return int(_native.load().tsde_rheun_mlp_lds(d, m, drift.hidden, diffusion.hidden, diffusion.out, noise, drift.n_mid,
                                                 diffusion.n_mid))


def forward(ys, z_out, y0, drift, diffusion, noise, m, schedule, times, bm, method=_native.TRAJ_REVERSIBLE_HEUN):
    """All steps of reversible Heun in one launch (``tsde_rheun_mlp_forward``): `ys` (n_out, rows, d) the outputs of
    `schedule`, `z_out` (rows, d) the scheme's second state after the last step; `times` (n_steps + 1) on the device.
    `method`: also Euler, midpoint, Heun, Euler-Heun for the same nets (``tsde_deep_mlp_forward``; `z_out`: the final state)."""
    rows, d = y0.shape
    _native.require_device(ys, y0, z_out, times)
    if ys.shape != (schedule.n_out, rows, d) or times.numel() != schedule.n_steps + 1:
        raise ValueError("shape mismatch: ys (n_out, rows, d), times (n_steps + 1)")
    if any(t.dtype != torch.float32 or not t.is_contiguous() for t in (ys, y0, z_out, times)):
        raise ValueError("the reversible-Heun kernels take contiguous float32 tensors")
    lib, dt_code, stream = K._launch_env(y0)
    entropy_dev = bm._entropy_dev
    fs, gs = drift.struct(), diffusion.struct()
    code = lib.tsde_deep_mlp_forward(ys.data_ptr(), z_out.data_ptr(), y0.data_ptr(), rows, d, int(m), int(noise),
                                     ctypes.byref(fs), ctypes.byref(gs), int(method), schedule.struct(), times.data_ptr(),
                                     bm._key, bm._elem0, None if entropy_dev is None else entropy_dev.data_ptr(), dt_code, stream)
    _native.check(code, "tsde_deep_mlp_forward")
    return ys


def _pad4(n):
    return (int(n) + 3) // 4 * 4


class _Stash:
    """The per-chunk buffers ``tsde_rheun_mlp_backward`` fills (tsde_rheun_stash_t)."""

    def __init__(self, n_eval, rows, d, m, drift, diffusion, general, device):
        self.sd, self.sm, self.shf, self.shg = _pad4(d), _pad4(m), _pad4(drift.hidden), _pad4(diffusion.hidden)

        def new(width):
            return torch.empty((n_eval, rows, width), dtype=torch.float32, device=device)
        self.z, self.cf, self.p = new(self.sd), new(self.sd), new(self.sd)
        self.q = new(self.sd) if general else None
        self.wa = new(self.sm) if general else None
        self.wb = new(self.sm) if general else None
        self.hf = [new(self.shf) for _ in range(drift.n_mid + 1)]
        self.df = [new(self.shf) for _ in range(drift.n_mid + 1)]
        self.hg = [new(self.shg) for _ in range(diffusion.n_mid + 1)]
        self.dg = [new(self.shg) for _ in range(diffusion.n_mid + 1)]
        s = _native.RheunStash()
        s.z, s.cf, s.p = self.z.data_ptr(), self.cf.data_ptr(), self.p.data_ptr()
        s.q = 0 if self.q is None else self.q.data_ptr()
        s.wa = 0 if self.wa is None else self.wa.data_ptr()
        s.wb = 0 if self.wb is None else self.wb.data_ptr()
        for l in range(MAX_MID + 1):
            s.hf[l] = self.hf[l].data_ptr() if l < len(self.hf) else 0
            s.df[l] = self.df[l].data_ptr() if l < len(self.df) else 0
            s.hg[l] = self.hg[l].data_ptr() if l < len(self.hg) else 0
            s.dg[l] = self.dg[l].data_ptr() if l < len(self.dg) else 0
        s.stride_d, s.stride_m, s.stride_hf, s.stride_hg = self.sd, self.sm, self.shf, self.shg
        self.struct = s

    @staticmethod
    def floats_per_row(d, m, drift, diffusion, general):
        return (_pad4(d) * (4 if general else 3) + (2 * _pad4(m) if general else 0)
                + 2 * (drift.n_mid + 1) * _pad4(drift.hidden) + 2 * (diffusion.n_mid + 1) * _pad4(diffusion.hidden))


def backward_chunk(state, stash, ys_all, grad_ys, drift, diffusion, noise, m, schedule, times, j_hi, j_lo, bm):
    """Evaluations j_hi ... j_lo of the backward sweep (``tsde_rheun_mlp_backward``); `state`: the six (rows, d) tensors
    (y, z, a_y, a_z, a_f, p), updated in place."""
    rows, d = state[0].shape
    lib, dt_code, stream = K._launch_env(state[0])
    st = _native.RheunState(*[t.data_ptr() for t in state])
    entropy_dev = bm._entropy_dev
    fs, gs = drift.struct(), diffusion.struct()
    code = lib.tsde_rheun_mlp_backward(ctypes.byref(st), ctypes.byref(stash.struct), ys_all.data_ptr(), grad_ys.data_ptr(), rows, d,
                                       int(m), int(noise), ctypes.byref(fs), ctypes.byref(gs), schedule.struct(),
                                       times.data_ptr(), int(j_hi), int(j_lo), bm._key, bm._elem0,
                                       None if entropy_dev is None else entropy_dev.data_ptr(), dt_code, stream)
    _native.check(code, "tsde_rheun_mlp_backward")


def _tall_product(a, b, m, n):
    """(a^T b)[:m, :n] and the column sums of a[:, :m] for tall (N, stride) stash views: tsde_gram_partials -- a product with a
    result of at most 64 x 64 and N in the millions, the shape the BLAS library serves worst (measured here: 120 of the 220 ms of
    a configs[2]-sized forward + backward went to its kernels before this call replaced them)."""
    w, sums = K.gram(a, b, column_sums=True)
    return w[:m, :n], sums[:m]


def _net_gradients(net, acc, z, t_eval, hid, delta, last_cot, d, last=True):
    """Adds one chunk's weight gradients of `net` to `acc` (a list in `net.parameters()` order). `z` (N, stride) the points of
    evaluation, `t_eval` (n,) their times, `hid[l]` / `delta[l]` (N, stride) the layers' activations / pre-activation
    cotangents, `last_cot` (N, stride) the cotangent of the last layer's output before `final` (`last=False`: that layer is
    somebody else's, `_general_last_layer`). n evaluations of N / n rows."""
    h = net.hidden
    n = t_eval.numel()
    g_w1, g_b1 = _tall_product(delta[0], z, h, d)                       # (hidden, d)
    if net.time_input:
        g_t = (delta[0].reshape(n, -1, delta[0].shape[1]).sum(dim=1)[:, :h] * t_eval.unsqueeze(1)).sum(dim=0)
        g_w1 = torch.cat([g_t.unsqueeze(1), g_w1], dim=1)
    grads = [(g_w1, g_b1)]
    for l in range(net.n_mid):
        grads.append(_tall_product(delta[l + 1], hid[l], h, h))
    if last:
        grads.append(_tall_product(last_cot, hid[net.n_mid], net.out, h))
    k = 0
    for (w, b), (gw, gb) in zip(net.linears, grads):
        acc[k] += gw
        k += 1
        if b is not None:
            acc[k] += gb
            k += 1


def _general_last_layer(net, acc_w, acc_b, hid, p, q, wa, wb, d, m):
    """The last layer of a general-noise diffusion net: the cotangent of its (rows, d, m) output is p (x) wa + q (x) wb (the
    increments already carry the net's `scale`), times final'(.) of the recomputed pre-activations -- formed, contracted with
    the hidden activations and summed over the stash rows by ``tsde_rheun_last_layer_grad`` (csrc/rheun_grad.hip) without ever
    materialising the (rows, d * m) cotangent. `hid`, `p`, `q`, `wa`, `wb`: (N, stride) views of the stash."""
    N = hid.shape[0]
    row_blocks = int(max(1, min(512, (N + 15) // 16)))
    gw = torch.empty((row_blocks, net.hidden, net.out), dtype=torch.float32, device=hid.device)
    gb = torch.empty((row_blocks, net.out), dtype=torch.float32, device=hid.device)
    lib, dt_code, stream = K._launch_env(hid)
    ns = net.struct()
    code = lib.tsde_rheun_last_layer_grad(gw.data_ptr(), gb.data_ptr(), hid.data_ptr(), p.data_ptr(), q.data_ptr(), wa.data_ptr(),
                                          wb.data_ptr(), N, int(d), int(m), ctypes.byref(ns), hid.shape[1], p.shape[1],
                                          wa.shape[1], row_blocks, dt_code, stream)
    _native.check(code, "tsde_rheun_last_layer_grad")
    acc_w += gw.sum(dim=0).t()
    if acc_b is not None:
        acc_b += gb.sum(dim=0)


def general_last_layer_reference(net, hid, p, q, wa, wb, d, m):
    """A torch statement of what `_general_last_layer` adds (tests): (dL/dW2 (out, hidden), dL/db2 (out,))."""
    h = net.hidden
    w2, b2 = net.linears[-1]
    top = hid[:, :h]
    cot = (p[:, :d].unsqueeze(2) * wa[:, :m].unsqueeze(1) + q[:, :d].unsqueeze(2) * wb[:, :m].unsqueeze(1))
    cot = cot.reshape(top.shape[0], d * m)
    if net.final != _native.FINAL_NONE:
        v = net.final_act(torch.nn.functional.linear(top, w2.detach(), None if b2 is None else b2.detach()))
        cot = cot * ((1.0 - v * v) if net.final == _native.FINAL_TANH else v * (1.0 - v))
    return cot.t() @ top, cot.sum(dim=0)


class ReversibleHeunFn(torch.autograd.Function):
    """ys (n_out + 1, rows, d) with a grad_fn towards y0 and every tensor of the two nets. Forward: one launch. Backward: the
    exact-gradient sweep of the reversible pair on the matrix cores, nothing of the trajectory stored."""

    @staticmethod
    def forward(ctx, drift, diffusion, noise, m, schedule, times, z_holder, bm, y0, *tensors):
        rows, d = y0.shape
        nf = len(drift.parameters())
        f_net, g_net = drift.rebuilt(tensors[:nf]), diffusion.rebuilt(tensors[nf:])
        y0c = _native.contiguous(y0.detach())
        if y0c.data_ptr() % 16:
            y0c = y0c.clone()
        ys = torch.empty((schedule.n_out + 1, rows, d), dtype=torch.float32, device=y0.device)
        ys[0].copy_(y0c)
        z_last = torch.empty_like(y0c)
        forward(ys[1:], z_last, y0c, f_net, g_net, noise, m, schedule, times, bm)
        ctx.save_for_backward(ys, z_last, *tensors)
        ctx.nets, ctx.noise, ctx.m, ctx.schedule, ctx.times, ctx.bm = (drift, diffusion), noise, m, schedule, times, bm
        if z_holder is not None:
            z_holder.append(z_last)        # (the scheme's second state after the last step: `extra=True` callers)
        return ys

    @staticmethod
    def backward(ctx, gys):
        ys, z_last, *tensors = ctx.saved_tensors
        drift, diffusion = ctx.nets
        nf = len(drift.parameters())
        f_net, g_net = drift.rebuilt(tensors[:nf]), diffusion.rebuilt(tensors[nf:])
        noise, m, schedule, times = ctx.noise, ctx.m, ctx.schedule, ctx.times
        rows, d = ys.shape[1], ys.shape[2]
        dev = ys.device
        general = noise == _native.NOISE_GENERAL
        gys = _native.contiguous(gys)
        K_steps = schedule.n_steps
        per_eval = rows * _Stash.floats_per_row(d, m, f_net, g_net, general) * 4
        chunk = int(max(1, min(K_steps + 1, STASH_BYTES // max(per_eval, 1))))
        stash = _Stash(chunk, rows, d, m, f_net, g_net, general, dev)
        zeros = lambda: torch.zeros((rows, d), dtype=torch.float32, device=dev)      # noqa: E731
        state = [ys[-1].clone(), z_last.clone(), gys[-1].clone(), zeros(), zeros(), zeros()]
        acc_f = [torch.zeros_like(t, dtype=torch.float32) for t in f_net.parameters()]
        acc_g = [torch.zeros_like(t, dtype=torch.float32) for t in g_net.parameters()]
        t_all = times
        j_hi = K_steps
        while j_hi >= 0:
            j_lo = max(0, j_hi - chunk + 1)
            n = j_hi - j_lo + 1
            backward_chunk(state, stash, ys, gys, f_net, g_net, noise, m, schedule, times, j_hi, j_lo, ctx.bm)
            t_eval = t_all[j_lo:j_hi + 1].flip(0)                           # evaluation e of the chunk is j = j_hi - e
            flat = lambda x: x[:n].reshape(n * rows, x.shape[2])           # noqa: E731
            z = flat(stash.z)
            _net_gradients(f_net, acc_f, z, t_eval, [flat(x) for x in stash.hf], [flat(x) for x in stash.df], flat(stash.cf), d)
            if general:
                # every layer but the last from the stash; the last one from (p, q) and the two increments
                k_last = len(acc_g) - (2 if g_net.linears[-1][1] is not None else 1)
                _net_gradients(g_net, acc_g, z, t_eval, [flat(x) for x in stash.hg], [flat(x) for x in stash.dg], None, d,
                               last=False)
                _general_last_layer(g_net, acc_g[k_last], acc_g[k_last + 1] if g_net.linears[-1][1] is not None else None,
                                    flat(stash.hg[g_net.n_mid]), flat(stash.p), flat(stash.q), flat(stash.wa), flat(stash.wb),
                                    d, m)
            else:
                _net_gradients(g_net, acc_g, z, t_eval, [flat(x) for x in stash.hg], [flat(x) for x in stash.dg], flat(stash.p), d)
            j_hi = j_lo - 1
        grad_y0 = state[2] if ctx.needs_input_grad[8] else None
        grads = [g.to(t.dtype) for g, t in zip(acc_f + acc_g, tensors)]
        grads = [g if need else None for g, need in zip(grads, ctx.needs_input_grad[9:])]
        return (None,) * 8 + (grad_y0, *grads)


_times_on_device = {}


def _device_times(times_host, device):
    """The step boundaries on the device, remembered by content (every iteration of a training loop has the same grid)."""
    host = np.ascontiguousarray(times_host, dtype=np.float32)
    key = (host.tobytes(), str(device))
    hit = _times_on_device.get(key)
    if hit is None:
        if len(_times_on_device) >= 16:
            _times_on_device.clear()
        hit = _times_on_device[key] = torch.from_numpy(host.copy()).to(device)
    return hit


def solve(y0, drift, diffusion, noise, m, schedule, times_host, bm, z_holder=None):
    """ys (n_out + 1, rows, d) of reversible Heun through the kernels, differentiable with respect to y0 and the nets' tensors.
    `times_host`: the n_steps + 1 step boundaries (numpy, in the state dtype). `z_holder`: a list that receives the scheme's
    second state after the last step."""
    times = _device_times(times_host, y0.device)
    tensors = drift.parameters() + diffusion.parameters()
    return ReversibleHeunFn.apply(drift, diffusion, int(noise), int(m), schedule, times, z_holder, bm, y0, *tensors)
