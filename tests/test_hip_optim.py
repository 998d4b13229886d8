"""nasseg_optim_step (csrc/optim.hip, engine/optim_native.py) against torch.optim + clip_grad_norm_ themselves:
the reference's end of a training step (src/engine/trainer.py:163-166,258-268; optimisers of
src/utils/solvers.py:6-52).  torch on the CPU is the reference here - fp32 for the comparison, the same sequence in
float64 to show how far fp32 itself is from the exact result."""
import copy

import numpy as np
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu

DEV = torch.device("cuda", 0) if torch.cuda.is_available() else None

SHAPES = [(1,), (7,), (4096,), (4097,), (3, 5, 7), (10000,), (32, 16, 3, 3), (8192,), (19,), (64, 33)]


def _params(seed, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return [nn.Parameter(torch.randn(*s, generator=g).to(dtype)) for s in SHAPES]


def _grads(ps, step, scale, dtype=torch.float32, skip=()):
    g = torch.Generator().manual_seed(1000 + step)
    out = []
    for i, p in enumerate(ps):
        v = (torch.randn(p.shape, generator=g) * scale).to(dtype)
        out.append(None if i in skip else v)
    return out


def _make(kind, params, hp):
    if kind == "sgd":
        return torch.optim.SGD(params, **hp)
    return torch.optim.Adam(params, **hp)


def _torch_sequence(kind_a, hp_a, kind_b, hp_b, clip_a, clip_b, steps, scale, dtype, skip=()):
    pa, pb = _params(1, dtype), _params(2, dtype)
    oa, ob = _make(kind_a, pa, hp_a), _make(kind_b, pb, hp_b)
    norms = []
    for s in range(steps):
        for ps, off in ((pa, 0), (pb, 50)):
            for p, g in zip(ps, _grads(ps, s + off, scale, dtype, skip)):
                p.grad = g
        n = []
        for ps, c in ((pa, clip_a), (pb, clip_b)):
            if c > 0:
                n.append(float(nn.utils.clip_grad_norm_(ps, c)))
        norms.append(n)
        oa.step()
        ob.step()
    return pa, pb, oa, ob, norms


CASES = [
    ("sgd", dict(lr=1e-3, momentum=0.9, weight_decay=1e-5), "adam", dict(lr=3e-3, weight_decay=1e-5), 3.0, 3.0, 0.05),
    ("sgd", dict(lr=1e-2, momentum=0.0, weight_decay=0.0), "adam", dict(lr=1e-3, betas=(0.8, 0.99), eps=1e-6), 0.0, 3.0, 1.0),
    ("adam", dict(lr=1e-3, weight_decay=1e-4), "sgd", dict(lr=1e-3, momentum=0.5), 1e3, 0.5, 0.2),  # no clipping / hard clipping
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_clip_and_step_against_torch_optim(case):
    from nas_segm_amd.engine.optim_native import NativeStep

    kind_a, hp_a, kind_b, hp_b, clip_a, clip_b, scale = CASES[case]
    steps = 6
    ra, rb, roa, rob, rnorms = _torch_sequence(kind_a, hp_a, kind_b, hp_b, clip_a, clip_b, steps, scale, torch.float32)
    da, db, doa, dob, _ = _torch_sequence(kind_a, hp_a, kind_b, hp_b, clip_a, clip_b, steps, scale, torch.float64)

    pa = [nn.Parameter(p.detach().to(DEV)) for p in _params(1)]
    pb = [nn.Parameter(p.detach().to(DEV)) for p in _params(2)]
    oa, ob = _make(kind_a, pa, hp_a), _make(kind_b, pb, hp_b)
    native = NativeStep.build([(pa, clip_a, oa), (pb, clip_b, ob)])
    assert native is not None
    for s in range(steps):
        for ps, off in ((pa, 0), (pb, 50)):
            for p, g in zip(ps, _grads(ps, s + off, scale)):
                p.grad = g.to(DEV)
        norms = native.step().cpu().tolist()
        assert np.allclose(norms[:len(rnorms[s])], rnorms[s], rtol=2e-6), (s, norms, rnorms[s])
        # gradients are clipped IN PLACE, as clip_grad_norm_ does
    # state with torch's names and types: what state_dict() / load_state_dict() exchange
    for o, ro in ((oa, roa), (ob, rob)):
        sd, rsd = o.state_dict(), ro.state_dict()
        assert sd["param_groups"] == rsd["param_groups"]
        assert sorted(sd["state"]) == sorted(rsd["state"])
        for k in rsd["state"]:
            assert sorted(sd["state"][k]) == sorted(rsd["state"][k]), k
            for name, v in rsd["state"][k].items():
                mine = sd["state"][k][name]
                if name == "step":
                    assert not mine.is_cuda and mine.dtype == v.dtype and float(mine) == float(v) == steps
    worst = 0.0
    for ps, rs, ds, o, ro, do in ((pa, ra, da, oa, roa, doa), (pb, rb, db, ob, rob, dob)):
        for i, (p, r, d) in enumerate(zip(ps, rs, ds)):
            floor = float((r.detach().double() - d.detach()).abs().max())  # fp32 torch against float64 torch
            err = float((p.detach().cpu().double() - d.detach()).abs().max())
            worst = max(worst, err / (floor + 1e-12))
            assert torch.allclose(p.detach().cpu(), r.detach(), rtol=2e-5, atol=1e-7), (i, err, floor)
            assert err <= 4 * floor + 1e-7, (i, err, floor)
            for name, v in ro.state[r].items():
                if name != "step":
                    assert torch.allclose(o.state[p][name].cpu(), v, rtol=2e-5,
                                          atol=2e-6 * float(v.abs().max())), (i, name)
    print("OPTIM case", case, "worst error / fp32-torch error vs float64:", round(worst, 2))


def test_clipped_gradients_are_written_back():
    from nas_segm_amd.engine.optim_native import NativeStep

    ps = [nn.Parameter(p.detach().to(DEV)) for p in _params(3)]
    # a parameter whose storage starts 4 bytes past a 16-byte boundary (a view into a flat buffer): scalar path
    flat = torch.randn(1001, generator=torch.Generator().manual_seed(9)).to(DEV)
    ps.append(nn.Parameter(flat[1:]))
    assert ps[-1].data_ptr() % 16 != 0 and ps[-1].is_contiguous()
    ref = [nn.Parameter(p.detach().cpu().clone()) for p in ps]
    o, ro = torch.optim.SGD(ps, lr=0.1, momentum=0.9), torch.optim.SGD(ref, lr=0.1, momentum=0.9)
    native = NativeStep.build([(ps, 0.7, o)])
    for s in range(3):
        gs = _grads(ref, s, 1.0)
        for p, r, g in zip(ps, ref, gs):
            p.grad, r.grad = g.to(DEV), g.clone()
        native.step()
        nn.utils.clip_grad_norm_(ref, 0.7)
        ro.step()
        for p, r in zip(ps, ref):
            assert torch.allclose(p.grad.cpu(), r.grad, rtol=2e-6, atol=1e-9)
            assert torch.allclose(p.detach().cpu(), r.detach(), rtol=1e-5, atol=1e-7)


def test_parameters_without_gradient_are_left_alone_and_torch_can_take_over():
    """a parameter whose grad is None: no state, no weight decay, no step count (torch skips it); then torch's own
    optim.step() on the state this left, then nasseg again: one sequence, as if torch had done every step"""
    from nas_segm_amd.engine.optim_native import NativeStep

    hp = dict(lr=3e-3, weight_decay=1e-5)
    skip = (2, 5)
    ref = _params(4)
    ro = torch.optim.Adam(ref, **hp)
    ps = [nn.Parameter(p.detach().to(DEV)) for p in _params(4)]
    o = torch.optim.Adam(ps, **hp)
    native = NativeStep.build([(ps, 3.0, o)])
    who = ["nasseg", "nasseg", "torch", "torch", "nasseg", "nasseg"]
    for s, by in enumerate(who):
        gs = _grads(ref, s, 0.3, skip=skip if s < 4 else ())
        for p, r, g in zip(ps, ref, gs):
            p.grad = None if g is None else g.to(DEV)
            r.grad = None if g is None else g.clone()
        nn.utils.clip_grad_norm_(ref, 3.0)
        ro.step()
        if by == "nasseg":
            native.step()
        else:
            nn.utils.clip_grad_norm_(ps, 3.0)
            o.step()
        if s == 1:
            for i in skip:
                assert ps[i] not in o.state or len(o.state[ps[i]]) == 0
    for i, (p, r) in enumerate(zip(ps, ref)):
        assert float(o.state[p]["step"]) == float(ro.state[r]["step"]) == (2 if i in skip else 6)
        assert torch.allclose(p.detach().cpu(), r.detach(), rtol=2e-5, atol=1e-7), i
    assert native.rebuilds >= 2  # (the set of live gradients changed)


def test_load_state_dict_between_steps():
    from nas_segm_amd.engine.optim_native import NativeStep

    hp = dict(lr=1e-3)
    ref = _params(5)
    ro = torch.optim.Adam(ref, **hp)
    ps = [nn.Parameter(p.detach().to(DEV)) for p in _params(5)]
    o = torch.optim.Adam(ps, **hp)
    native = NativeStep.build([(ps, 0.0, o)])
    saved = None
    for s in range(5):
        for p, r, g in zip(ps, ref, _grads(ref, s, 0.3)):
            p.grad, r.grad = g.to(DEV), g.clone()
        native.step()
        ro.step()
        if s == 1:
            saved, rsaved = copy.deepcopy(o.state_dict()), copy.deepcopy(ro.state_dict())
        if s == 3:  # back to the moments and the step count of step 1 (bias correction restarts from 2)
            o.load_state_dict(saved)
            ro.load_state_dict(rsaved)
    for p, r in zip(ps, ref):
        assert float(o.state[p]["step"]) == float(ro.state[r]["step"]) == 3
        assert torch.allclose(p.detach().cpu(), r.detach(), rtol=2e-5, atol=1e-7)


def test_what_is_not_plain_sgd_or_adam_goes_to_torch():
    from nas_segm_amd.engine.optim_native import NativeStep
    from nas_segm_amd.engine.trainer_common import clip_and_step

    ps = [nn.Parameter(p.detach().to(DEV)) for p in _params(6)]
    assert NativeStep.build([(ps, 1.0, torch.optim.SGD(ps, lr=0.1, nesterov=True, momentum=0.9))]) is None
    assert NativeStep.build([(ps, 1.0, torch.optim.Adam(ps, lr=0.1, amsgrad=True))]) is None
    assert NativeStep.build([(ps, 1.0, torch.optim.Adam(ps, lr=0.1, capturable=True))]) is None
    assert NativeStep.build([(ps, 1.0, torch.optim.AdamW(ps, lr=0.1))]) is None
    assert NativeStep.build([(ps, 1.0, torch.optim.RMSprop(ps, lr=0.1))]) is None
    assert NativeStep.build([(ps[:3], 1.0, torch.optim.SGD(ps, lr=0.1))]) is None  # clip set != stepped set
    # ... and clip_and_step still steps them
    o = torch.optim.RMSprop(ps, lr=0.1)
    before = [p.detach().clone() for p in ps]
    for p, g in zip(ps, _grads(ps, 0, 1.0)):
        p.grad = g.to(DEV)
    clip_and_step([(ps, 1.0, o)])
    assert all(not torch.equal(b, p.detach()) for b, p in zip(before, ps))


def test_against_torch_optimisers_on_the_device():
    """torch's own kernels on the same device: plain SGD and SGD with weight decay give identical bits (the engine
    tests compare steps taken by either); with momentum 1 element in a thousand differs in its last bit, Adam's second
    moment likewise (where torch's compiler fused a multiply-add is its own matter - tools/diag_optim_bits.py)"""
    from nas_segm_amd.engine.optim_native import NativeStep

    for kind, hp, exact in (("sgd", dict(lr=1e-3), True), ("sgd", dict(lr=1e-3, weight_decay=1e-5), True),
                            ("sgd", dict(lr=1e-3, momentum=0.9, weight_decay=1e-5), False),
                            ("adam", dict(lr=3e-3, weight_decay=1e-5), False)):
        ref = [nn.Parameter(p.detach().to(DEV)) for p in _params(7)]
        ps = [nn.Parameter(p.detach().to(DEV)) for p in _params(7)]
        ro, o = _make(kind, ref, hp), _make(kind, ps, hp)
        native = NativeStep.build([(ps, 0.0, o)])
        for s in range(4):
            for p, r, g in zip(ps, ref, _grads(ref, s, 0.3)):
                p.grad, r.grad = g.to(DEV), g.to(DEV)
            native.step()
            ro.step()
        for p, r in zip(ps, ref):
            if exact:
                assert torch.equal(p.detach(), r.detach()), hp
            else:
                assert torch.allclose(p.detach(), r.detach(), rtol=1e-6, atol=1e-8), hp


def test_cached_stepper_follows_the_objects_it_is_handed():
    """clip_and_step caches its NativeStep on the optimiser; another parameter list, an added param_group or another
    clip norm must not be stepped with the old tables"""
    from nas_segm_amd.engine.trainer_common import clip_and_step

    ps = [nn.Parameter(p.detach().to(DEV)) for p in _params(8)]
    extra = nn.Parameter(torch.randn(33, device=DEV))
    ref = [nn.Parameter(p.detach().cpu().clone()) for p in ps]
    rextra = nn.Parameter(extra.detach().cpu().clone())
    o, ro = torch.optim.SGD(ps, lr=0.1, momentum=0.9), torch.optim.SGD(ref, lr=0.1, momentum=0.9)
    for s in range(4):
        if s == 2:
            o.add_param_group({"params": [extra], "lr": 0.05})
            ro.add_param_group({"params": [rextra], "lr": 0.05})
            ps, ref = ps + [extra], ref + [rextra]
        for p, r, g in zip(ps, ref, _grads(ref, s, 1.0)):
            p.grad, r.grad = g.to(DEV), g.clone()
        clip = 0.7 if s < 3 else 0.4
        clip_and_step([(ps, clip, o)])
        nn.utils.clip_grad_norm_(ref, clip)
        ro.step()
        from nas_segm_amd.engine.optim_native import cached_stepper
        assert cached_stepper(o) is not None
    for p, r in zip(ps, ref):
        assert torch.allclose(p.detach().cpu(), r.detach(), rtol=1e-5, atol=1e-7)


def test_cache_dies_with_its_optimiser():
    """the stepper cache is keyed weakly by the first optimiser and its values hold the optimisers weakly: dropping
    optimiser and parameters frees the stepper - parameters, gradients, state, device tables and pinned buffers (the
    search makes new optimisers per candidate and task, src/main_search.py:575)"""
    import gc
    import weakref

    from nas_segm_amd.engine import optim_native
    from nas_segm_amd.engine.trainer_common import clip_and_step

    gc.collect()
    before = len(optim_native._CACHE)
    probes = []
    for seed in range(5):
        ps = [nn.Parameter(p.detach().to(DEV)) for p in _params(20 + seed)]
        o = torch.optim.Adam(ps, lr=1e-3)
        for p, g in zip(ps, _grads(ps, 0, 1.0)):
            p.grad = g.to(DEV)
        clip_and_step([(ps, 1.0, o)])
        stepper = optim_native.cached_stepper(o)
        assert stepper is not None
        probes.append((weakref.ref(o), weakref.ref(stepper), weakref.ref(ps[0])))
        del ps, o, stepper, p, g
    gc.collect()
    assert len(optim_native._CACHE) == before
    assert all(r() is None for probe in probes for r in probe)


def test_adam_parameter_that_misses_steps_keeps_its_own_step_count():
    """a parameter without a gradient is not stepped - by torch or here - and its bias correction resumes from ITS
    count when the gradient comes back (the device counters advance for stepped rows only)"""
    from nas_segm_amd.engine.optim_native import NativeStep

    ref = [nn.Parameter(p.detach().clone()) for p in _params(11)]
    ps = [nn.Parameter(p.detach().to(DEV)) for p in _params(11)]
    ro, o = torch.optim.Adam(ref, lr=3e-3), torch.optim.Adam(ps, lr=3e-3)
    native = NativeStep.build([(ps, 0.0, o)])
    for s in range(6):
        skip = (1, 4) if 1 <= s <= 3 else ()
        for p, r, g in zip(ps, ref, _grads(ref, s, 0.5, skip=skip)):
            p.grad, r.grad = (None if g is None else g.to(DEV)), g
        native.step()
        ro.step()
    for i, (p, r) in enumerate(zip(ps, ref)):
        assert float(o.state[p]["step"]) == float(ro.state[r]["step"]), i
        assert torch.allclose(p.detach().cpu(), r.detach(), rtol=2e-5, atol=1e-7), i
    assert [float(v) for v in native.dstep.cpu()] == [float(ro.state[r]["step"]) for r in ref]
