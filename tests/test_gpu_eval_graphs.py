"""The decoder module's host paths for planned calls.  Evaluation-shaped calls (no gradient, planned pair budget):
DecoderSplattingCUDA.forward at the reference's test_step shape (b = 1, v = 3, model_wrapper.py:415-454) is launch-bound
from Python; the module runs such calls on forward-only PREPARED steps (keyed by shapes, inputs bound per call) and keeps
round 5's cache of captured HIP graphs (second call of the same addresses captured, later ones one graph launch) for plans
that cannot be prepared -- the cache's own tests below switch the prepared steps off.  Training calls: prepared steps with
one autograd node, from the compiled binding or from Python (decoder.py)."""
import pytest
import torch

from spfsplatv2_amd import synthetic as syn
from tests import util

pytestmark = pytest.mark.gpu


def _setup(seed=9, prepare=False, **kw):
    import spfsplatv2_amd as spf
    from spfsplatv2_amd import decoder as dec
    kw = {**dict(s_mult=4.0, G=3000, K=25, image_hw=(96, 80)), **kw}
    b = syn.make_batch("TEST", 1, 3, seed=seed, **kw).to("cuda")
    d = util.product_decoder()
    d.prepare_steps = prepare         # (False: the graph cache's own tests -- with prepared steps on, evaluation calls of a
    #                                    plan with direct bins never reach it)
    g = dec.Gaussians(b.means, b.covariances, b.rotations, b.scales, b.harmonics, b.opacities)
    args = lambda gg=g, bb=b: (gg, bb.extrinsics, bb.intrinsics, bb.near, bb.far, bb.image_shape)
    with torch.no_grad():
        exact = d.forward(*args())                                   # exact mode: no plan, no graph
    assert not d._graphs
    d.max_pairs = spf.plan_pair_budget(d.last_call, check="deferred")
    return spf, dec, b, d, g, args, exact


def test_second_call_is_captured_and_results_are_the_callers(hip_lib):
    spf, dec, b, d, g, args, exact = _setup()
    with torch.no_grad():
        o1 = d.forward(*args())                                      # first sight of the key: launched as usual
        assert not d._graphs
        o2 = d.forward(*args())                                      # second: captured, then replayed
        assert len(d._graphs) == 1
        o3 = d.forward(*args())                                      # replay
        out4, alpha4, radii4 = d.render(*args())                     # replay, all four outputs
    assert len(d._graphs) == 1 and spf.plan_flags(d.last_call) == 0
    for o in (o1, o2, o3, out4):
        assert torch.equal(o.color, exact.color) and torch.equal(o.depth, exact.depth)
    # nothing a call returned aliases anything a later call returned or the graph's own buffers
    ptrs = {o.color.data_ptr() for o in (o1, o2, o3, out4)} | {o.depth.data_ptr() for o in (o1, o2, o3, out4)}
    assert len(ptrs) == 8
    held = o2.color.clone()
    with torch.no_grad():
        b.extrinsics[0, 0, 0, 3] += 0.05                             # same address, new content: the replay must see it
        moved = d.forward(*args())
        d.eval_graphs = False
        moved_eager = d.forward(*args())
        d.eval_graphs = True
    assert torch.equal(o2.color, held)                               # the earlier result did not change under the caller
    assert torch.equal(moved.color, moved_eager.color) and not torch.equal(moved.color, exact.color)
    assert float(alpha4.min()) >= 0.0 and radii4.dtype == torch.int32 and radii4.shape == (1, 3, g.means.shape[1])


def test_new_addresses_and_gradients_never_replay(hip_lib):
    spf, dec, b, d, g, args, exact = _setup(seed=10)
    with torch.no_grad():
        d.forward(*args()); d.forward(*args())
        assert len(d._graphs) == 1
        g2 = dec.Gaussians(*(t.clone() for t in (g.means, g.covariances, g.rotations, g.scales, g.harmonics, g.opacities)))
        g2.opacities.mul_(0.5)
        other = d.forward(*args(g2))                                 # other tensors: another key, launched as usual
        d.eval_graphs = False
        want = d.forward(*args(g2))
        d.eval_graphs = True
    assert len(d._graphs) == 1 and torch.equal(other.color, want.color)
    # a call that will be differentiated never comes near the cache
    ext = b.extrinsics.clone().requires_grad_(True)
    out = d.forward(g, ext, b.intrinsics, b.near, b.far, b.image_shape)
    out.color.mean().backward()
    assert ext.grad is not None and bool(torch.isfinite(ext.grad).all()) and len(d._graphs) == 1
    d.clear_eval_graphs()
    assert not d._graphs


def test_replayed_call_whose_plan_fails_is_rerun_in_exact_mode(hip_lib):
    """check="backward" plans are verified after the replay (one host read); a plan that does not hold for the inputs
    of THIS call -- here: bins of 2 entries -- gives the exact-mode result, not NaN."""
    spf, dec, b, d, g, args, exact = _setup(seed=11)
    d.max_pairs = d.max_pairs._replace(check="backward")
    with torch.no_grad():
        d.forward(*args()); d.forward(*args())
        ok = d.forward(*args())
        assert torch.equal(ok.color, exact.color)
        d.clear_eval_graphs()
        d.max_pairs = d.max_pairs._replace(max_tile_list=2, check="deferred")
        d.forward(*args()); bad = d.forward(*args())
        assert spf.plan_flags(d.last_call) & 2 and len(d._graphs) == 1   # deferred: NaN images, flag readable
        assert bool(torch.isnan(bad.color).all())
        d.clear_eval_graphs()
        d.max_pairs = d.max_pairs._replace(check="backward")
        with pytest.raises(spf._lib.SpfError):
            d.forward(*args())                                       # first sight, launched as usual: raises as ever
        fixed = d.forward(*args())                                   # second: captured, replayed, verified, re-run exactly
        assert torch.equal(fixed.color, exact.color)
        again = d.forward(*args())                                   # and every later replay is verified just the same
        assert torch.equal(again.color, exact.color)


def test_a_general_call_between_replays_leaves_the_graphs_record_alone(hip_lib):
    """`last_call` after a replay is the GRAPH's record (its counters and verdict word are what the next replay is
    checked by).  A general-path call in between -- here an exact-mode one -- writes the decoder's own record, not that
    one: the next replay, on inputs that outgrew the plan at the same addresses, is still caught and re-run exactly."""
    spf, dec, b, d, g, args, exact = _setup(seed=12)
    plan = d.max_pairs._replace(check="backward")
    d.max_pairs = plan
    with torch.no_grad():
        d.forward(*args()); d.forward(*args())
        assert len(d._graphs) == 1 and torch.equal(d.forward(*args()).color, exact.color)
        graph_record = d.last_call
        d.max_pairs = None
        assert torch.equal(d.forward(*args()).color, exact.color)    # exact mode, the general launcher
        assert d.last_call is not graph_record and graph_record.get("counters") is not None
        d.max_pairs = plan
        b.scales.mul_(150.0)                                         # same addresses, footprints x 150: the plan fails
        want = util.product_decoder().forward(*args())
        got = d.forward(*args())                                     # replayed, verified against ITS OWN record, re-run
    assert len(d._graphs) == 1 and not bool(torch.isnan(got.color).any()) and torch.equal(got.color, want.color)
    assert not torch.equal(got.color, exact.color)


def _auto_decoder(slack=1.3, defer=False):
    d = util.product_decoder(auto_plan=slack)
    d.auto_plan_defer = defer
    return d


@pytest.mark.parametrize("prepare", [True, False], ids=["prepared_steps", "graph_cache"])
def test_auto_plan_evaluation_never_returns_a_failed_plan(hip_lib, prepare):
    """decoder.auto_plan (on by default): the module plans for itself.  First call of a shape: exact mode; the next ones
    planned (and, being evaluation calls, run on a forward-only prepared step -- or, with those off, captured); inputs
    that outgrow the plan are re-run in exact mode at once and re-planned."""
    import spfsplatv2_amd as spf
    from spfsplatv2_amd import decoder as dec
    small = syn.make_batch("TEST", 1, 3, seed=21, s_mult=2.0, G=3000, K=4, image_hw=(96, 80)).to("cuda")
    big = syn.make_batch("TEST", 1, 3, seed=21, s_mult=40.0, G=3000, K=4, image_hw=(96, 80)).to("cuda")
    gs = lambda b: dec.Gaussians(b.means, b.covariances, b.rotations, b.scales, b.harmonics, b.opacities)
    call = lambda d, b, g: d.forward(g, b.extrinsics, b.intrinsics, b.near, b.far, b.image_shape)
    ref, d = util.product_decoder(), _auto_decoder()
    d.prepare_steps = prepare
    g_small, g_big = gs(small), gs(big)
    with torch.no_grad():
        want_small, want_big = call(ref, small, g_small), call(ref, big, g_big)
        first = call(d, small, g_small)
        assert isinstance(d.max_pairs, spf.PairBudget) and d.last_call.get("counters") is None       # ran exact, planned
        plan = d.max_pairs
        outs = [call(d, small, g_small) for _ in range(3)]                                           # planned; captured; replayed
        assert d.max_pairs.capacity == plan.capacity
        assert (len(d._graphs), sum(map(len, d._prepared_steps.values()))) == ((0, 1) if prepare else (1, 0))
        for o in [first] + outs:
            assert torch.equal(o.color, want_small.color) and torch.equal(o.depth, want_small.depth)
        grown = call(d, big, g_big)                                   # same shape, many times the pairs: the plan cannot hold
        assert torch.equal(grown.color, want_big.color) and d.max_pairs.capacity > plan.capacity
        after = [call(d, big, g_big) for _ in range(3)]
        for o in after:
            assert torch.equal(o.color, want_big.color)
        assert spf.plan_flags(d.last_call) == 0


def test_default_decoder_plans_for_itself_and_always_returns_exact_results(hip_lib):
    """A decoder nobody configured (what an unchanged caller of the reference gets): the first call of a shape is exact,
    later TRAINING calls are planned and verified when their forward has been issued; one that outgrew its plan is re-run
    exactly on the spot -- images bit-equal to exact mode, gradients to rounding, never NaN, nothing raised.  A `max_pairs`
    set by the caller switches the module's own planning off."""
    import spfsplatv2_amd as spf
    from spfsplatv2_amd import decoder as dec
    small = syn.make_batch("TEST", 1, 3, seed=24, s_mult=2.0, G=3000, K=4, image_hw=(96, 80)).to("cuda")
    big = syn.make_batch("TEST", 1, 3, seed=24, s_mult=300.0, G=3000, K=4, image_hw=(96, 80)).to("cuda")

    def step(d, b):
        means = b.means.clone().requires_grad_(True)
        g = dec.Gaussians(means, b.covariances, b.rotations, b.scales, b.harmonics, b.opacities)
        out = d.forward(g, b.extrinsics, b.intrinsics, b.near, b.far, b.image_shape)
        out.color.square().mean().backward()
        return out.color.detach(), means.grad

    ref = util.product_decoder()                                       # exact mode, every call
    d = dec.get_decoder(dec.DecoderSplattingCUDACfg(name="splatting_cuda", background_color=[0.0, 0.0, 0.0],
                                                    make_scale_invariant=True, enable_cov_grad=True,
                                                    enable_sh_grad=True)).cuda()
    assert d.auto_plan == 1.5 and not d.auto_plan_defer and d.max_pairs is None
    want_small, want_big = step(ref, small), step(ref, big)
    for i in range(3):
        c, g = step(d, small)
        assert torch.equal(c, want_small[0]) and util.rel_linf(g, want_small[1]) < 1e-5
        assert isinstance(d.max_pairs, spf.PairBudget) and (d.last_call.get("counters") is None) == (i == 0)
    cap = d.max_pairs.capacity
    for i in range(2):
        c, g = step(d, big)                                           # i = 0: plan fails, re-run exactly, re-planned
        assert torch.equal(c, want_big[0]) and util.rel_linf(g, want_big[1]) < 1e-5
    assert d.max_pairs.capacity > cap and d.last_call.get("counters") is not None
    # the caller's own plan switches the module's planning off
    mine = d.max_pairs._replace(check="deferred")
    d.max_pairs = mine
    step(d, small)
    assert d.max_pairs is mine and spf.plan_flags(d.last_call) == 0
    d.max_pairs = None
    c, g = step(d, small)                                             # back to the module: the shape's plan is remembered
    assert d.last_call.get("counters") is not None and torch.equal(c, want_small[0])
    # two shapes alternating (training views / one validation view): each keeps its plan, only the first call of each is exact
    one = syn.make_batch("TEST", 1, 1, seed=25, s_mult=2.0, G=3000, K=4, image_hw=(96, 80)).to("cuda")
    want_one = step(ref, one)
    exact_calls = 0
    for i in range(3):
        for b, want in ((one, want_one), (small, want_small)):
            c, g = step(d, b)
            exact_calls += d.last_call.get("counters") is None
            assert torch.equal(c, want[0]) and util.rel_linf(g, want[1]) < 1e-5
    assert exact_calls == 1, exact_calls


def test_auto_plan_deferred_training_reads_the_verdict_one_call_late(hip_lib):
    """`auto_plan_defer` (opt-in): a training call under an automatic plan waits for nothing, its verdict is read at the next
    call.  A step whose plan failed is all NaN (the reference's NaN-gradient guard skips it, model_wrapper.py:1117-1151),
    the next one runs in exact mode and re-plans."""
    import spfsplatv2_amd as spf
    from spfsplatv2_amd import decoder as dec
    small = syn.make_batch("TEST", 1, 3, seed=22, s_mult=2.0, G=3000, K=4, image_hw=(96, 80)).to("cuda")
    big = syn.make_batch("TEST", 1, 3, seed=22, s_mult=40.0, G=3000, K=4, image_hw=(96, 80)).to("cuda")

    def step(d, b):
        means = b.means.clone().requires_grad_(True)
        g = dec.Gaussians(means, b.covariances, b.rotations, b.scales, b.harmonics, b.opacities)
        out = d.forward(g, b.extrinsics, b.intrinsics, b.near, b.far, b.image_shape)
        out.color.square().mean().backward()
        return out.color.detach(), means.grad

    ref, d = util.product_decoder(), _auto_decoder(defer=True)
    want_small, want_big = step(ref, small), step(ref, big)
    c, g = step(d, small)                                             # exact, plans
    assert torch.equal(c, want_small[0]) and isinstance(d.max_pairs, spf.PairBudget)
    for _ in range(2):                                                # planned: same images, gradients to rounding
        c, g = step(d, small)
        assert torch.equal(c, want_small[0]) and util.rel_linf(g, want_small[1]) < 1e-5
        assert d._auto_pending is not None and not d._graphs
    cap = d.max_pairs.capacity
    c, g = step(d, big)                                               # outgrows the plan: NaN everywhere, nothing raised
    assert bool(torch.isnan(c).all()) and bool(torch.isnan(g).all())
    c, g = step(d, big)                                               # verdict read: exact mode again, new plan
    assert torch.equal(c, want_big[0]) and util.rel_linf(g, want_big[1]) < 1e-5 and d.max_pairs.capacity > cap
    c, g = step(d, big)                                               # planned under the new plan
    assert torch.equal(c, want_big[0]) and spf.plan_flags(d.last_call) == 0


# ---- training calls: the module's prepared steps (round 6) ----------------------------------------------------------
def _train_setup(seed=31, split=False):
    import spfsplatv2_amd as spf
    from spfsplatv2_amd import decoder as dec
    b = syn.make_batch("TEST", 2, 2, seed=seed, s_mult=4.0, G=2500, K=25, image_hw=(96, 80)).to("cuda")
    leaves = {n: getattr(b, n).clone().requires_grad_(True) for n in util.GRAD_NAMES}
    high = None
    if split:
        high = b.harmonics[..., 16:].contiguous().requires_grad_(True)
        leaves["harmonics"] = b.harmonics[..., :16].contiguous().requires_grad_(True)
    g = dec.Gaussians(leaves["means"], None, leaves["rotations"], leaves["scales"], leaves["harmonics"], leaves["opacities"],
                      harmonics_band4=high)
    probe = util.product_decoder()
    with torch.no_grad():
        probe.forward(g, leaves["extrinsics"], b.intrinsics, b.near, b.far, b.image_shape)
    plan = spf.plan_pair_budget(probe.last_call, slack=1.3, check="deferred")
    w = torch.rand(2, 2, 3, 96, 80, device="cuda", generator=torch.Generator("cuda").manual_seed(5))

    def step(d, with_depth=False, retain=False):
        for t in leaves.values():
            t.grad = None
        out = d.forward(g, leaves["extrinsics"], b.intrinsics, b.near, b.far, b.image_shape)
        loss = (out.color * w).sum() + (0.1 * (out.depth * w[:, :, 0]).sum() if with_depth else 0.0)
        loss.backward(retain_graph=retain)
        return out, loss, {n: t.grad.clone() for n, t in leaves.items()}

    return spf, b, leaves, g, plan, step, w


@pytest.fixture(params=["compiled_step", "python_step"])
def step_binding(request, monkeypatch):
    """Prepared steps run from the compiled binding (csrc/torch_binding.cpp::PreparedStep) when it has been built and
    from rasterizer.StaticStep's own ctypes calls otherwise: both under test."""
    from spfsplatv2_amd import _lib
    if request.param == "python_step":
        monkeypatch.setattr(_lib, "_fast", None)
    elif _lib.fast() is None or not hasattr(_lib.fast(), "PreparedStep"):
        pytest.skip("the compiled binding has not been built")
    return request.param


def _same(a, b):
    return torch.equal(a[0].color, b[0].color) and torch.equal(a[0].depth, b[0].depth) and \
        all(torch.equal(a[2][n], b[2][n]) for n in a[2])


@pytest.mark.parametrize("split", [False, True], ids=["dense_sh", "split_sh"])
def test_training_calls_replay_from_graphs_bit_identically(hip_lib, split, step_binding):
    """VERDICT r5 task 5: a planned training call whose shapes repeat runs on a PREPARED step (static state, argument
    structs built once: five C-ABI calls) -- images, depth and every gradient bit-identical to the general path; what a
    call returned stays the caller's; an in-place update of an input (an optimizer step) is seen; a depth gradient and a
    retained graph's second backward give the same numbers, and a forward issued before the previous backward gets a
    prepared step of its own."""
    spf, b, leaves, g, plan, step, w = _train_setup(split=split)
    eager, d = util.product_decoder(max_pairs=plan), util.product_decoder(max_pairs=plan)
    eager.prepare_steps = False
    want = step(eager)
    first = step(d)
    assert not d._prepared_steps and _same(first, want)                 # first sight of the key: launched as usual
    second = step(d)
    assert len(d._prepared_steps) == 1 and _same(second, want)          # captured, replayed
    held = second[0].color.clone()
    third = step(d)
    assert _same(third, want) and torch.equal(second[0].color, held) and third[0].color.data_ptr() != second[0].color.data_ptr()
    assert all((e.step.fast is not None) == (step_binding == "compiled_step") for es in d._prepared_steps.values() for e in es)
    assert spf.plan_flags(d.last_call) == 0 and eager._prepared_steps == {}
    # an optimizer step: same addresses, new values
    with torch.no_grad():
        leaves["means"].add_(0.01 * torch.randn_like(leaves["means"]))
        leaves["extrinsics"][:, :, 0, 3] += 0.02
    moved = step(d)
    assert _same(moved, step(eager)) and not torch.equal(moved[0].color, want[0].color)
    # a gradient of the depth output: not what the backward graph was captured for -> eager kernels, same state
    assert _same(step(d, with_depth=True), step(eager, with_depth=True))
    # a retained graph: its second backward runs eagerly into fresh buffers
    out, loss, g1 = step(d, retain=True)
    for t in leaves.values():
        t.grad = None
    loss.backward()
    assert all(torch.equal(leaves[n].grad, g1[n]) for n in leaves)
    # a forward before the previous one's backward: the second call gets a step of its own, both are right
    for t in leaves.values():
        t.grad = None
    o1 = d.forward(g, leaves["extrinsics"], b.intrinsics, b.near, b.far, b.image_shape)
    o2 = d.forward(g, leaves["extrinsics"], b.intrinsics, b.near, b.far, b.image_shape)
    ((o1.color * w).sum() + (o2.color * w).sum()).backward()
    both = {n: t.grad.clone() for n, t in leaves.items()}
    ref = step(eager)
    assert torch.equal(o1.color, o2.color) and torch.equal(o1.color, ref[0].color)
    assert all(util.rel_linf(both[n], 2 * ref[2][n]) < 1e-6 for n in leaves)
    assert len(d._prepared_steps) == 1 and [len(v) for v in d._prepared_steps.values()] == [2]
    # ... and a third one waiting at the same time: the general path (two steps per shape), still right
    for t in leaves.values():
        t.grad = None
    outs = [d.forward(g, leaves["extrinsics"], b.intrinsics, b.near, b.far, b.image_shape) for _ in range(3)]
    sum((o.color * w).sum() for o in outs).backward()
    assert all(torch.equal(o.color, ref[0].color) for o in outs) and [len(v) for v in d._prepared_steps.values()] == [2]
    assert all(util.rel_linf(leaves[n].grad, 3 * ref[2][n]) < 1e-6 for n in leaves)
    d.clear_prepared_steps()
    assert _same(step(d), step(eager))


def test_prepared_steps_follow_fresh_tensors_of_every_call(hip_lib, step_binding):
    """What an encoder hands the decoder: NEW tensors every step (other addresses, other values).  The prepared step is
    keyed by shapes and binds its inputs per call, so the second call prepares it and every later one runs on it --
    bit-identical to the general path on the same values; the tensors of a call outlive its backward (the step holds
    them), and nothing of an older call is read."""
    spf, b, leaves, g, plan, step, w = _train_setup(seed=37)
    from spfsplatv2_amd import decoder as dec
    eager, d = util.product_decoder(max_pairs=plan), util.product_decoder(max_pairs=plan)
    eager.prepare_steps = False
    seen = set()
    for i in range(5):
        fresh = {n: (t.detach() + (0.003 * i if n == "means" else 0.0)).clone().requires_grad_(True) for n, t in leaves.items()}
        cams = [t.clone() for t in (b.intrinsics, b.near, b.far)]
        seen.add(fresh["means"].data_ptr())
        res = []
        for m in (d, eager):
            # (means and opacities as an encoder hands them over: non-leaf tensors with a graph behind them)
            gi = dec.Gaussians(fresh["means"] * 1.0, None, fresh["rotations"], fresh["scales"], fresh["harmonics"],
                               fresh["opacities"] + 0.0)
            for t in fresh.values():
                t.grad = None
            out = m.forward(gi, fresh["extrinsics"], *cams, b.image_shape)
            loss = (out.color * w).sum() + 0.1 * (out.depth * w[:, :, 0]).sum()
            del gi
            if m is d and i == 3:
                # the caller drops everything but the loss: the inputs must still be there for the backward's kernels
                held = {n: t for n, t in fresh.items()}
                grads = torch.autograd.grad(loss, list(held.values()))
                res.append((out.color.clone(), out.depth.clone(), {n: gr for n, gr in zip(held, grads)}))
                continue
            loss.backward()
            res.append((out.color.clone(), out.depth.clone(), {n: t.grad.clone() for n, t in fresh.items()}))
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]), i
        assert all(torch.equal(res[0][2][n], res[1][2][n]) for n in fresh), i
    entries = [e for es in d._prepared_steps.values() for e in es]
    assert len(entries) == 1 and entries[0].gen == 4 and len(seen) > 1      # call 0 the general way, calls 1 - 4 on ONE step
    assert eager._prepared_steps == {}


def test_evaluation_calls_whose_tensors_move_run_on_a_forward_only_step(hip_lib, step_binding):
    """A validation loop hands the decoder NEW tensors every batch: the graph cache (keyed by addresses) never sees a key
    twice.  Such calls run on a forward-only prepared step (keyed by shapes, inputs bound per call): results are exact
    mode's bit for bit, follow the tensors of the call, and a plan that does not hold for a call's inputs is caught behind
    the projection kernel and the call re-run exactly -- never NaN."""
    spf, dec, b, d, g, args, exact = _setup(seed=13, prepare=True)
    d.max_pairs = d.max_pairs._replace(check="backward")
    ref = util.product_decoder()

    def fresh(scale_opacity=1.0, scale_scales=1.0):
        gg = dec.Gaussians(b.means.clone(), None, b.rotations.clone(), b.scales * scale_scales, b.harmonics.clone(),
                           b.opacities * scale_opacity)
        return gg, b.extrinsics.clone(), b.intrinsics.clone(), b.near.clone(), b.far.clone(), b.image_shape

    with torch.no_grad():
        outs = [d.forward(*fresh()) for _ in range(4)]              # general, prepared, bound, bound
        entries = [e for es in d._prepared_steps.values() for e in es]
        assert len(entries) == 1 and entries[0].step.gpair is None and not d._graphs
        assert (entries[0].step.fast is not None) == (step_binding == "compiled_step")
        assert all(torch.equal(o.color, exact.color) and torch.equal(o.depth, exact.depth) for o in outs)
        assert len({o.color.data_ptr() for o in outs}) == 4
        dim = fresh(scale_opacity=0.5)
        got, want = d.render(*dim), ref.render(*dim)
        assert torch.equal(got[0].color, want[0].color) and torch.equal(got[1], want[1]) and torch.equal(got[2], want[2])
        assert not torch.equal(got[0].color, exact.color) and spf.plan_flags(d.last_call) == 0
        big = fresh(scale_scales=150.0)                              # the plan cannot hold for these
        got, want = d.forward(*big), ref.forward(*big)
        assert not bool(torch.isnan(got.color).any()) and torch.equal(got.color, want.color)
        again = d.forward(*fresh())                                  # and the step is as good as before
        assert torch.equal(again.color, exact.color)
        # tensors that stay where they are run on the same step; the graph cache has nothing to do
        for _ in range(3):
            still = d.forward(*args())
        assert not d._graphs and torch.equal(still.color, exact.color) and entries[0].gen == 0
        assert [len(v) for v in d._prepared_steps.values()] == [1]


def test_prepared_step_backward_into_a_gradient_bucket(hip_lib, step_binding):
    """Data-parallel ranks run the backward inside `with shard.GradBucket(...)`: the prepared step then hands the general
    backward its state and the CURRENT binding's inputs (also when the forward ran in the compiled step) -- gradients
    land in the bucket's views, equal to the unbucketed ones bit for bit."""
    from spfsplatv2_amd import shard
    spf, b, leaves, g, plan, step, w = _train_setup(seed=41)
    d = util.product_decoder(max_pairs=plan)
    step(d); want = step(d)                                              # second call: prepared
    assert len(d._prepared_steps) == 1
    for t in leaves.values():
        t.grad = None
    out = d.forward(g, leaves["extrinsics"], b.intrinsics, b.near, b.far, b.image_shape)
    loss = (out.color * w).sum()
    bucket = shard.GradBucket(*(leaves[n] for n in ("means", "scales", "rotations", "opacities", "harmonics")))
    with bucket:
        loss.backward()
    lo, hi = bucket.flat.data_ptr(), bucket.flat.data_ptr() + 4 * bucket.flat.numel()
    for n in ("means", "scales", "rotations", "opacities", "harmonics"):
        assert lo <= leaves[n].grad.data_ptr() < hi and torch.equal(leaves[n].grad, want[2][n]), n
    assert torch.equal(leaves["extrinsics"].grad, want[2]["extrinsics"]) and torch.equal(out.color, want[0].color)


def test_training_graph_with_a_plan_that_fails_is_rerun_exactly(hip_lib, step_binding):
    """The module's own planning over prepared training calls: inputs that outgrow the plan AT THE SAME ADDRESSES (the
    prepared step exists) -> the forward's early check raises inside the module, the call is re-run in exact mode,
    re-planned and prepared again under the new plan."""
    from spfsplatv2_amd import decoder as dec
    small = syn.make_batch("TEST", 1, 3, seed=26, s_mult=2.0, G=3000, K=4, image_hw=(96, 80)).to("cuda")
    big = syn.make_batch("TEST", 1, 3, seed=26, s_mult=300.0, G=3000, K=4, image_hw=(96, 80)).to("cuda")
    means, scales = small.means.clone().requires_grad_(True), small.scales.clone()
    g = dec.Gaussians(means, None, small.rotations, scales, small.harmonics, small.opacities)

    def step(d):
        means.grad = None
        out = d.forward(g, small.extrinsics, small.intrinsics, small.near, small.far, small.image_shape)
        out.color.square().mean().backward()
        return out.color.detach().clone(), means.grad.clone()

    ref, d = util.product_decoder(), _auto_decoder(slack=1.5)
    want = step(ref)
    for i in range(4):                                               # exact, planned, captured, replayed
        c, gr = step(d)
        assert torch.equal(c, want[0]) and util.rel_linf(gr, want[1]) < 1e-5
    assert len(d._prepared_steps) == 1
    cap = d.max_pairs.capacity
    with torch.no_grad():
        scales.copy_(big.scales)                                     # same address, footprints x 150
    want_big = step(ref)
    for i in range(4):
        c, gr = step(d)
        assert torch.equal(c, want_big[0]) and util.rel_linf(gr, want_big[1]) < 1e-5, i
    assert d.max_pairs.capacity > cap and len(d._prepared_steps) >= 1
