"""Adam as the reference configures it (run_plnerf.py:446-447: lr, betas (0.9, 0.999), eps 1e-8, no weight
decay, no amsgrad), stepping a whole parameter group with ONE launch of plnerf_adam_step.

`FlatAdam` is a torch.optim.Adam: same constructor arguments, same `param_groups` (the training loop rewrites
`group['lr']` every iteration, run_plnerf.py:1311-1315), same `state_dict()` layout -- so the checkpoint the
reference writes (run_plnerf.py:1324-1332, `optimizer_state_dict`) loads here and vice versa.  What differs is
the storage: at construction the group's parameters are re-homed as consecutive slices of one flat fp32 buffer
(their `.data` become views; `load_state_dict` / `copy_` keep working), and so are `exp_avg` / `exp_avg_sq`.
functional.MlpFn.backward hands autograd the network's 24 gradients as slices of one buffer in the same order,
so a step is one elementwise kernel over four flat arrays per network instead of a multi-tensor launch over 24
tensor lists; gradients in any other layout are stepped run by run (worst case one launch per tensor).
Parameters without a gradient are skipped, as in torch.  CPU parameters fall back to torch's own step.
"""
import torch

from . import _lib as L


def contiguous_runs(tensors):
    """Split a list of fp32 tensors into maximal runs that are consecutive slices, in order, of one buffer.
    Returns [(first index, one-past-last index, flat view over the run)]; a tensor that is not contiguous fp32
    forms a run of its own with view None."""
    runs = []
    i, n = 0, len(tensors)
    while i < n:
        t0 = tensors[i]
        if t0.dtype != torch.float32 or not t0.is_contiguous():
            runs.append((i, i + 1, None))
            i += 1
            continue
        store_end = t0.untyped_storage().data_ptr() + t0.untyped_storage().nbytes()
        expect = t0.data_ptr() + t0.numel() * 4
        total = t0.numel()
        j = i + 1
        while j < n:
            t = tensors[j]
            if (t.dtype != torch.float32 or not t.is_contiguous() or t.device != t0.device or t.data_ptr() != expect
                    or expect + t.numel() * 4 > store_end):     # (neighbouring allocations that merely touch)
                break
            expect += t.numel() * 4
            total += t.numel()
            j += 1
        runs.append((i, j, t0.as_strided((total,), (1,))))
        i = j
    return runs


def flat_view_of(tensors):
    """One flat view over `tensors` if they are consecutive slices, in order, of one buffer; else None."""
    if not tensors or any(t is None for t in tensors):
        return None
    runs = contiguous_runs(tensors)
    return runs[0][2] if len(runs) == 1 else None


class FlatAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, guards=None):
        """guards: optional list of 1-element int32 device tensors (NeRF.status_word()); while any of them is non-zero
        the step kernels leave parameters and moments untouched (a step whose forward left the half range must not
        reach the weights; the host finds out at its next NeRF.check_range())."""
        super().__init__(params, lr=lr, betas=betas, eps=eps)
        self.guards = list(guards or [])
        self._flat = []          # per group: dict(param, m, v, grad, step) or None (torch's own step)
        for group in self.param_groups:
            ps = group['params']
            ok = len(ps) > 0 and all(p.is_cuda and p.dtype == torch.float32 and p.device == ps[0].device for p in ps)
            self._flat.append(self._flatten(ps) if ok else None)

    def _flatten(self, ps):
        sizes = [p.numel() for p in ps]
        dev = ps[0].device
        flat = torch.empty(sum(sizes), device=dev, dtype=torch.float32)
        m = torch.zeros_like(flat)
        v = torch.zeros_like(flat)
        for p, fp, fm, fv in zip(ps, flat.split(sizes), m.split(sizes), v.split(sizes)):
            fp = fp.view(p.shape)
            fp.copy_(p.data)
            p.data = fp
            # torch's own state layout, so that state_dict() is the reference's optimizer_state_dict
            self.state[p] = {'step': torch.tensor(0.0), 'exp_avg': fm.view(p.shape), 'exp_avg_sq': fv.view(p.shape)}
        return {'param': flat, 'm': m, 'v': v, 'sizes': sizes}

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)      # replaces the state tensors: move them back into the flat buffers
        for group, fl in zip(self.param_groups, self._flat):
            if fl is None:
                continue
            for p, fm, fv in zip(group['params'], fl['m'].split(fl['sizes']), fl['v'].split(fl['sizes'])):
                st = self.state.get(p)
                if not st:
                    self.state[p] = {'step': torch.tensor(0.0), 'exp_avg': fm.view(p.shape).zero_(),
                                     'exp_avg_sq': fv.view(p.shape).zero_()}
                    continue
                fm.view(p.shape).copy_(st['exp_avg'])
                fv.view(p.shape).copy_(st['exp_avg_sq'])
                st['exp_avg'], st['exp_avg_sq'] = fm.view(p.shape), fv.view(p.shape)
                st['step'] = torch.as_tensor(float(st['step']), dtype=torch.float32)

    def _guard_ptr(self, first, last):
        """One status word can guard a launch.  With several guards (one Adam over both networks, the depth variant)
        they are OR-ed into a scratch word first."""
        if not self.guards:
            return None
        if len(self.guards) == 1:
            return L.dptr(self.guards[0], "guard", torch.int32)
        if getattr(self, "_guard_any", None) is None:
            self._guard_any = torch.zeros(1, device=self.guards[0].device, dtype=torch.int32)
        self._guard_any.copy_(torch.stack([g.reshape(()) for g in self.guards]).amax().reshape(1))
        return L.dptr(self._guard_any, "guard", torch.int32)

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if any(fl is None for fl in self._flat):
            if grad_scale != 1.0:
                raise RuntimeError("FlatAdam: grad_scale needs flat (GPU fp32) parameter groups")
            super().step(closure=None)      # (the closure, if any, was evaluated above)
            return loss
        for group, fl in zip(self.param_groups, self._flat):
            ps = group['params']
            if all(p.grad is None for p in ps):
                continue
            if group.get('weight_decay', 0) or group.get('amsgrad') or group.get('maximize'):
                raise RuntimeError("FlatAdam implements the reference's Adam: no weight decay / amsgrad / maximize")
            view = flat_view_of([p.data for p in ps])
            if view is None or view.data_ptr() != fl['param'].data_ptr():
                # not this optimizer's buffer any more: .to(device) after construction, or a second FlatAdam built over
                # the same parameters re-homed them into ITS buffer -- stepping fl['param'] would update dead memory
                raise RuntimeError("FlatAdam: the parameters no longer live in this optimizer's flat buffer (they were "
                                   "re-allocated, or another FlatAdam was built over them afterwards); build one "
                                   "optimizer per parameter set, after .to(device)")
            # parameters without a gradient are skipped, as torch.optim.Adam does; the others are stepped run by
            # run: a run = consecutive parameters whose gradients are consecutive slices of one buffer and whose
            # step counts agree (one network's backward = one run)
            offs = [0]
            for n in fl['sizes']:
                offs.append(offs[-1] + n)
            live = [k for k, p in enumerate(ps) if p.grad is not None]
            b1, b2 = group['betas']
            k = 0
            while k < len(live):
                e = k + 1
                while e < len(live) and live[e] == live[e - 1] + 1 and \
                        float(self.state[ps[live[e]]]['step']) == float(self.state[ps[live[k]]]['step']):
                    e += 1
                idx = live[k:e]
                for a, b, gview in contiguous_runs([ps[q].grad for q in idx]):
                    first, last = idx[a], idx[b - 1]
                    if gview is None:
                        gview = ps[first].grad.to(torch.float32).contiguous().reshape(-1)
                    lo, hi = offs[first], offs[last + 1]
                    steps = [self.state[ps[q]]['step'] for q in idx[a:b]]
                    torch._foreach_add_(steps, 1.0)
                    L.check(L.lib().plnerf_adam_step(
                        L.dptr(fl['param'][lo:hi]), L.dptr(gview), L.dptr(fl['m'][lo:hi]), L.dptr(fl['v'][lo:hi]),
                        hi - lo, float(group['lr']), float(b1), float(b2), float(group['eps']),
                        int(steps[0].item()), float(grad_scale), self._guard_ptr(first, last), L.stream()),
                        "plnerf_adam_step")
                k = e
        return loss
