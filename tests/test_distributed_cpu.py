"""world_size-2 gloo tests (CPU) of the data-parallel plumbing: flat-bucket
gradient all-reduce, parameter broadcast and confusion-matrix reduction."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nas_segm_amd.engine import RankParallel

        torch.manual_seed(100 + rank)  # different init per rank: broadcast must fix it
        net = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.BatchNorm1d(3), torch.nn.Linear(3, 2))
        net.register_parameter("never_reached", torch.nn.Parameter(torch.ones(3)))  # e.g. unused aux head
        dp = RankParallel(net)
        first = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
        gathered = [torch.zeros_like(first) for _ in range(world)]
        dist.all_gather(gathered, first)
        same_params = all(torch.equal(gathered[0], g) for g in gathered)

        # autograd writes fresh gradients; sync packs them into the flat bucket (one multi-tensor
        # copy), all-reduces it and leaves every reached param.grad pointing into the bucket
        x = torch.full((5, 4), float(rank + 1))
        assert dp.attach_flat_grads() is None and all(p.grad is None for p in net.parameters())
        net(x).sum().backward()
        reached = [p for p in net.parameters() if p is not net.never_reached]
        local = [p.grad.clone() for p in reached]
        dp.sync_gradients()
        # fixed layout: every trainable parameter (reached by this loss or not) + one status element
        assert dp._flat.numel() == sum(p.numel() for p in net.parameters()) + 1
        views = dict((id(p), v) for p, v in zip(dp._parameters_once(), dp._views))
        assert all(p.grad.data_ptr() == views[id(p)].data_ptr() for p in reached)
        assert float(views[id(net.never_reached)].abs().sum()) == 0.0
        value, peers_ok = dp.step_values(torch.tensor(1.5))
        assert value == 1.5 and peers_ok
        ok_alias = net.never_reached.grad is None
        for p, l in zip(reached, local):
            allg = [torch.zeros_like(l) for _ in range(world)]
            dist.all_gather(allg, l)
            ok_alias &= torch.allclose(p.grad, sum(allg) / world, atol=1e-6)
        # path 2: grads allocated by autograd after zero_grad(set_to_none)
        for p in net.parameters():
            p.grad = None
        net(x * 2).sum().backward()
        local = [p.grad.clone() for p in reached]
        dp.sync_gradients()
        ok_pack = net.never_reached.grad is None
        for p, l in zip(reached, local):
            allg = [torch.zeros_like(l) for _ in range(world)]
            dist.all_gather(allg, l)
            ok_pack &= torch.allclose(p.grad, sum(allg) / world, atol=1e-6)
        cm = torch.full((3, 3), rank + 1, dtype=torch.int64)
        dp.reduce_confusion(cm)
        ok_cm = bool((cm == sum(range(1, world + 1))).all())
        q.put((rank, same_params, ok_alias, ok_pack, ok_cm, dp.world_size))
    finally:
        dist.destroy_process_group()


def test_rank_parallel_two_processes_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same_params, ok_alias, ok_pack, ok_cm, ws in results:
        assert same_params and ok_alias and ok_pack and ok_cm and ws == 2, results


def test_rank_parallel_single_process_is_a_noop():
    from nas_segm_amd.engine import RankParallel

    net = torch.nn.Linear(3, 2)
    dp = RankParallel(net)
    assert dp.world_size == 1 and dp.module is net
    net(torch.ones(1, 3)).sum().backward()
    g = net.weight.grad.clone()
    dp.sync_gradients()
    assert torch.equal(net.weight.grad, g)


def _search_worker(rank, world, port, q, log_path):
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nas_segm_amd.engine.search import search_loop

        counter = {"n": 0}
        trained = []

        def sample_fn():  # only ever called on rank 0 (the controller lives there)
            assert rank == 0
            counter["n"] += 1
            n = counter["n"]
            return [[n, n + 1, 0], [[0, 1, n % 3, 0, 1]]], 0.5 * n, -1.0 * n

        def evaluate_fn(config):  # stands in for evaluate_candidate: reward depends on the genotype
            return 0.01 * config[0][0] + 0.001 * rank, 1000 + config[0][0]

        writer = open(log_path, "w") if rank == 0 else None
        hist = search_loop(sample_fn, trained.append, evaluate_fn, 2, arch_writer=writer, first_epoch=5)
        if writer:
            writer.close()
        q.put((rank, hist, trained))
    finally:
        dist.destroy_process_group()


def test_search_loop_one_candidate_per_rank_gloo(tmp_path):
    """config-4 outer loop: rank 0 samples `world` genotypes per iteration, every rank evaluates
    its own, rank 0 trains the controller once per candidate in sampling order and writes the
    reference's genotype log format"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    log_path = str(tmp_path / "genotypes.out")
    procs = [ctx.Process(target=_search_worker, args=(r, world, port, q, log_path)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict((r[0], r[1:]) for r in [q.get(timeout=120) for _ in range(world)])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    hist, trained = results[0]
    assert results[1] == (None, [])
    assert [c[0][0] for c, _ in hist] == [1, 2, 3, 4]
    # candidate k of an iteration ran on rank k: reward = 0.01*n + 0.001*rank
    assert [round(r, 4) for _, r in hist] == [0.01, 0.021, 0.03, 0.041]
    assert [(t[0][0][0], round(t[1], 4), t[2], t[3]) for t in trained] == [
        (1, 0.01, 0.5, -1.0), (2, 0.021, 1.0, -2.0), (3, 0.03, 1.5, -3.0), (4, 0.041, 2.0, -4.0)]
    lines = open(log_path).read().strip().split("\n")
    assert len(lines) == 4
    assert lines[0].startswith("reward: 0.0100, epoch: 5, params: 1001, epoch_time: ")
    assert lines[3].startswith("reward: 0.0410, epoch: 8, params: 1004, epoch_time: ")
    assert lines[1].endswith("genotype: [[2, 3, 0], [[0, 1, 2, 0, 1]]]")


def test_search_loop_single_process():
    from nas_segm_amd.engine.search import search_loop

    seen = []
    hist = search_loop(lambda: ([[1]], 0.1, -0.2), seen.append, lambda cfg: 0.5, 3)
    assert hist == [([[1]], 0.5)] * 3 and seen == [([[1]], 0.5, 0.1, -0.2)] * 3


# ---------------------------------------------------------------------------
# the engine's entry points on two ranks.  The CPU has no product kernels, so the handful of
# functional ops the engine itself calls are replaced by their torch equivalents inside the
# worker and the "segmenter" is a two-layer torch net behind the product's Segmenter /
# RankParallel - what is under test is the rank protocol, not the kernels.
# ---------------------------------------------------------------------------
def _torch_functional():
    import torch.nn.functional as TF

    from nas_segm_amd import functional as F

    def nearest_label_resize(t, size, out=None):
        y = TF.interpolate(t[:, None].float(), size=tuple(size), mode="nearest").long()[:, 0]
        return y if out is None else out.copy_(y)

    F.nearest_label_resize = nearest_label_resize
    F.copy_into = lambda dst, src: dst.copy_(src)
    F.gather_rows = lambda src, idx, out=None: src[idx]
    F.log_softmax_nll = lambda logits, target, ignore_index=255: TF.nll_loss(
        TF.log_softmax(logits, 1), target, ignore_index=ignore_index)
    F.bilinear_resize = lambda x, size: x if tuple(x.shape[2:]) == tuple(size) else TF.interpolate(
        x, size=tuple(size), mode="bilinear", align_corners=False)

    def argmax_confusion(logits, gt, n, cm=None, **kw):
        pred = TF.interpolate(logits, size=tuple(gt.shape[1:]), mode="bilinear", align_corners=False).argmax(1)
        keep = gt < n
        if cm is None:
            cm = torch.zeros((n, n), dtype=torch.int64)
        cm += torch.bincount(gt[keep].long() * n + pred[keep], minlength=n * n).view(n, n)
        return cm

    F.argmax_confusion = argmax_confusion


class _ToyEncoder(torch.nn.Module):
    def __init__(self, fail_at=None):
        super(_ToyEncoder, self).__init__()
        self.conv = torch.nn.Conv2d(3, 4, 3, stride=2, padding=1)
        self.bn = torch.nn.BatchNorm2d(4)
        self.calls, self.fail_at = 0, fail_at

    def forward(self, x):
        self.calls += 1
        if self.fail_at is not None and self.calls == self.fail_at:
            raise RuntimeError("HIP out of memory (simulated)")
        return [torch.relu(self.bn(self.conv(x)))]


class _ToyDecoder(torch.nn.Module):
    def __init__(self):
        super(_ToyDecoder, self).__init__()
        self.clf = torch.nn.Conv2d(4, 5, 1)

    def forward(self, feats):
        return self.clf(feats[0])


class _Crit(object):
    ignore_index = 255


def _toy_batches(rank, n, seed=0):
    g = torch.Generator().manual_seed(1000 * seed + rank)
    return [{"image": torch.randn(2, 3, 8, 12, generator=g),
             "mask": torch.randint(0, 5, (2, 8, 12), generator=g).to(torch.uint8)} for _ in range(n)]


def _engine_worker(rank, world, port, q):
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nas_segm_amd.engine import RankParallel, Segmenter
        from nas_segm_amd.engine.inference import validate
        from nas_segm_amd.engine.trainer import populate_task0, train_segmenter, train_task0

        _torch_functional()
        out = {}

        def candidate(fail_at=None):
            torch.manual_seed(7)
            net = Segmenter(_ToyEncoder(fail_at), _ToyDecoder())
            dp = RankParallel(net)
            oe = torch.optim.SGD(net.encoder.parameters(), lr=0.1, momentum=0.9)
            od = torch.optim.Adam(net.decoder.parameters(), lr=0.01)
            return net, dp, oe, od

        def same_on_all_ranks(module):
            flat = torch.cat([p.detach().reshape(-1) for p in module.parameters()])
            allv = [torch.zeros_like(flat) for _ in range(world)]
            dist.all_gather(allv, flat)
            return all(torch.equal(allv[0], v) for v in allv)

        # (1) a healthy epoch: parameters stay identical on all ranks (averaged gradients)
        net, dp, oe, od = candidate()
        init = torch.cat([p.detach().reshape(-1).clone() for p in net.parameters()])
        ret = train_segmenter(dp, _toy_batches(rank, 3), oe, od, 0, _Crit(), False, 3.0, 3.0, False)
        out["healthy"] = (ret, same_on_all_ranks(net),
                          bool((torch.cat([p.detach().reshape(-1) for p in net.parameters()]) != init).any()))
        r = validate(dp, _toy_batches(rank, 2, seed=1), 0, 0, num_classes=5, omit_classes=[])
        rs = [None] * world
        dist.all_gather_object(rs, float(r))
        out["reward_same"] = all(abs(v - rs[0]) < 1e-12 for v in rs) and 0.0 < rs[0] <= 1.0

        # (2) rank 1 fails in its SECOND step: every rank must leave train_segmenter with 0 at that
        # step (nobody left waiting in an all-reduce), and the collectives that follow still pair
        net, dp, oe, od = candidate(fail_at=2 if rank == 1 else None)
        ret = train_segmenter(dp, _toy_batches(rank, 4), oe, od, 0, _Crit(), False, 3.0, 3.0, False)
        probe = torch.tensor([float(rank + 1)])
        dist.all_reduce(probe)
        out["train_failure"] = (ret, net.encoder.calls, float(probe))

        # (3) rank 0 fails during validation
        net, dp, oe, od = candidate(fail_at=2 if rank == 0 else None)
        r = validate(dp, _toy_batches(rank, 3, seed=2), 0, 0, num_classes=5, omit_classes=[])
        probe = torch.tensor([float(rank + 1)])
        dist.all_reduce(probe)
        out["val_failure"] = (r, float(probe))

        # (4) task0 on a SHARDED feature cache: every rank caches its own samples, steps on its
        # own batches, the decoder gradients are averaged - decoders stay identical
        net, dp, oe, od = candidate()
        shard = [{"image": b["image"][i:i + 1], "mask": b["mask"][i:i + 1]}
                 for b in _toy_batches(rank, 2, seed=3) for i in range(2)]
        Xy = populate_task0(dp, shard, None, 4, do_kd=False)
        before = torch.cat([p.detach().reshape(-1).clone() for p in net.decoder.parameters()])
        ret = train_task0(Xy, dp, od, 0, _Crit(), None, 2, False, False, 0.0, 3.0, False)
        after = torch.cat([p.detach().reshape(-1) for p in net.decoder.parameters()])
        feats = [None] * world
        dist.all_gather_object(feats, float(Xy[0].double().abs().sum()))
        out["task0"] = (ret, int(Xy[0].shape[0]), tuple(Xy["y"].shape), same_on_all_ranks(net.decoder),
                        bool((after != before).any()), feats[0] != feats[1])
        # ... and a rank failing in its task0 step takes the others with it
        net, dp, oe, od = candidate()
        if rank == 1:
            net.decoder.forward = lambda feats: (_ for _ in ()).throw(RuntimeError("simulated"))
        ret = train_task0(Xy, dp, od, 0, _Crit(), None, 2, False, False, 0.0, 3.0, False)
        probe = torch.tensor([float(rank + 1)])
        dist.all_reduce(probe)
        out["task0_failure"] = (ret, float(probe))

        # (5) round 3: the protocol covers the WHOLE step, and the ranks agree on their step counts.
        # (a) loaders of unequal length: both ranks run min(len) = 2 steps, nobody waits for a third
        net, dp, oe, od = candidate()
        ret = train_segmenter(dp, _toy_batches(rank, 3 if rank == 0 else 2), oe, od, 0, _Crit(), False, 3.0, 3.0, False)
        probe = torch.tensor([float(rank + 1)])
        dist.all_reduce(probe)
        out["uneven_loaders"] = (ret, net.encoder.calls, same_on_all_ranks(net), float(probe))

        # (b) the LOADER of rank 1 raises while fetching its second batch (not inside forward / backward)
        class FlakyLoader(list):
            def __iter__(self):
                for i, b in enumerate(list.__iter__(self)):
                    if rank == 1 and i == 1:
                        raise RuntimeError("DataLoader worker died (simulated)")
                    yield b

        net, dp, oe, od = candidate()
        ret = train_segmenter(dp, FlakyLoader(_toy_batches(rank, 3)), oe, od, 0, _Crit(), False, 3.0, 3.0, False)
        probe = torch.tensor([float(rank + 1)])
        dist.all_reduce(probe)
        out["loader_failure"] = (ret, net.encoder.calls, float(probe))

        # (c) the optimiser of rank 0 raises AFTER the gradient all-reduce of step 1: the flag travels with
        # the next step's collective, every rank stops there
        net, dp, oe, od = candidate()
        if rank == 0:
            real_step, n_calls = od.step, [0]

            def flaky_step(*a, **k):
                n_calls[0] += 1
                if n_calls[0] == 1:
                    raise RuntimeError("optimizer state allocation failed (simulated)")
                return real_step(*a, **k)

            od.step = flaky_step
        ret = train_segmenter(dp, _toy_batches(rank, 4), oe, od, 0, _Crit(), False, 3.0, 3.0, False)
        probe = torch.tensor([float(rank + 1)])
        dist.all_reduce(probe)
        out["optimiser_failure"] = (ret, net.encoder.calls, float(probe))

        # (d) cache shards of unequal size: train_task0 agrees on the smaller one's number of passes
        net, dp, oe, od = candidate()
        n_mine = 4 if rank == 0 else 2
        shard = [{"image": b["image"][i:i + 1], "mask": b["mask"][i:i + 1]}
                 for b in _toy_batches(rank, 2, seed=4) for i in range(2)][:n_mine]
        Xy = populate_task0(dp, shard, None, n_mine, do_kd=False)
        ret = train_task0(Xy, dp, od, 0, _Crit(), None, 2, False, False, 0.0, 3.0, False)
        probe = torch.tensor([float(rank + 1)])
        dist.all_reduce(probe)
        out["uneven_shards"] = (ret, int(Xy[0].shape[0]), same_on_all_ranks(net.decoder), float(probe))

        # (6) round 4.  (a) a failure that is NOT a RuntimeError, in the epoch's LAST step, AFTER that step's
        # gradient collective (rank 0's optimiser raises IndexError in step 3 of 3): rank 0 re-raises it as
        # RankFailure (try_except scores 0 instead of letting the process die), tells its peers through the
        # end-of-epoch handshake, and they score 0 too - nobody goes on to validation alone
        net, dp, oe, od = candidate()
        if rank == 0:
            real_step, n_calls = od.step, [0]

            def late_step(*a, **k):
                n_calls[0] += 1
                if n_calls[0] == 3:
                    raise IndexError("index out of range in optimiser state (simulated)")
                return real_step(*a, **k)

            od.step = late_step
        ret = train_segmenter(dp, _toy_batches(rank, 3), oe, od, 0, _Crit(), False, 3.0, 3.0, False)
        probe = torch.tensor([float(rank + 1)])
        dist.all_reduce(probe)
        out["last_step_failure"] = (ret, net.encoder.calls, float(probe))

        # (b) replayed steps take part in the SAME protocol as host-launched ones.  The CPU has no hipGraph:
        # the stepper's graph is replaced by an object whose replay() runs forward + backward on the host -
        # everything after the replay (RankParallel.sync_gradients with its status element and step count,
        # clipping, the optimisers) is the product's code.  Rank 1's loader dies before its second step while
        # rank 0 replays: both leave with 0 at that step.
        from nas_segm_amd.engine import graphed, trainer

        class HostReplayStep(graphed.GraphedSegmenterStep):
            built = 0

            def _capture(self, warmup):
                HostReplayStep.built += 1
                if getattr(self.segmenter, "_refuse_capture", False):
                    raise RuntimeError("HIP out of memory while capturing (simulated)")
                stepper, seg, model = self, self.segmenter, self.model

                class Graph(object):
                    def replay(self):
                        stepper.segmenter, stepper.model = seg, model
                        try:
                            stepper.loss = stepper._fwd_bwd(False)
                        finally:
                            stepper.segmenter = stepper.model = None
                        stepper._static_grads = [(p, p.grad) for p in stepper._params if p.grad is not None]

                self.graph = Graph()
                self._static_grads = []

        real_replays, real_step_cls = trainer._replays, graphed.GraphedSegmenterStep
        trainer._replays = lambda segmenter, device, n_pixels: True
        graphed.GraphedSegmenterStep = HostReplayStep
        try:
            net, dp, oe, od = candidate()
            ret = train_segmenter(dp, _toy_batches(rank, 3), oe, od, 0, _Crit(), False, 3.0, 3.0, False)
            out["replayed_healthy"] = (ret, HostReplayStep.built, dp.sync_count, same_on_all_ranks(net))
            net, dp, oe, od = candidate()
            ret = train_segmenter(dp, FlakyLoader(_toy_batches(rank, 3)), oe, od, 0, _Crit(), False, 3.0, 3.0, False)
            probe = torch.tensor([float(rank + 1)])
            dist.all_reduce(probe)
            out["replayed_loader_failure"] = (ret, net.encoder.calls, dp.sync_count, float(probe))
            # (c) the capture fails on rank 1 only: it launches from the host while rank 0 replays - one bucket
            # layout, so the collectives pair and the replicas stay identical
            net, dp, oe, od = candidate()
            dp._refuse_capture = rank == 1
            built = HostReplayStep.built
            ret = train_segmenter(dp, _toy_batches(rank, 3), oe, od, 0, _Crit(), False, 3.0, 3.0, False)
            out["mixed_replay_and_host"] = (ret, HostReplayStep.built - built, dp.sync_count, same_on_all_ranks(net))
        finally:
            trainer._replays, graphed.GraphedSegmenterStep = real_replays, real_step_cls
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_engine_entry_points_two_ranks_gloo():
    """train_segmenter / validate / populate_task0 / train_task0 on two ranks: replicas stay in
    step, the task0 cache is sharded, and a RuntimeError on ONE rank scores the candidate 0 on ALL
    ranks at the same step instead of leaving the others in an all-reduce (the reference's
    try_except convention carried over to one process per GPU)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_engine_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in range(world):
        out = results[rank]
        assert out["healthy"] == (None, True, True), out["healthy"]
        assert out["reward_same"]
        ret, calls, probe = out["train_failure"]
        assert ret == 0 and calls == 2 and probe == 3.0, out["train_failure"]  # both stopped at step 2
        assert out["val_failure"] == (0, 3.0), out["val_failure"]
        ret, n, yshape, same, moved, shards_differ = out["task0"]
        assert ret is None and n == 4 and yshape == (4, 4, 6) and same and moved and shards_differ, out["task0"]
        assert out["task0_failure"] == (0, 3.0), out["task0_failure"]
        assert out["uneven_loaders"] == (None, 2, True, 3.0), out["uneven_loaders"]
        ret, calls, probe = out["loader_failure"]
        assert ret == 0 and probe == 3.0 and calls in (1, 2), out["loader_failure"]  # (rank 1 never ran step 2)
        ret, calls, probe = out["optimiser_failure"]
        assert ret == 0 and probe == 3.0 and calls <= 2, out["optimiser_failure"]
        ret, n, same, probe = out["uneven_shards"]
        assert ret is None and n == (4 if rank == 0 else 2) and same and probe == 3.0, out["uneven_shards"]
        assert out["last_step_failure"] == (0, 3, 3.0), out["last_step_failure"]
        assert out["replayed_healthy"] == (None, 1, 3, True), out["replayed_healthy"]
        ret, calls, syncs, probe = out["replayed_loader_failure"]
        assert ret == 0 and probe == 3.0 and calls in (1, 2) and syncs == 2, out["replayed_loader_failure"]
        assert out["mixed_replay_and_host"] == (None, 1, 3, True), out["mixed_replay_and_host"]


def _ctrl_search_worker(rank, world, port, q, log_path):
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from _util import load_json
        from nas_segm_amd.engine.search import search_loop

        samples = load_json("controller.json")["wacv"]["samples"]
        it = iter(samples)
        inserted = []

        def sample_fn():
            s = next(it)
            return s["config"], s["entropy"], s["log_prob"]

        def evaluate_fn(config):  # rewards 0.01, 0.02, ... in sampling order, whatever rank evaluates
            k = [s["config"] for s in samples].index(config)
            return 0.01 * (k + 1), 1000 + k

        writer = open(log_path, "w") if rank == 0 else None
        search_loop(sample_fn, inserted.append, evaluate_fn, len(samples) // world, arch_writer=writer)
        if writer:
            writer.close()
        q.put((rank, inserted))
    finally:
        dist.destroy_process_group()


def test_search_loop_feeds_the_controller_like_the_reference_gloo(tmp_path):
    """config-4 outer loop with the records the REFERENCE controller sampled (golden): two
    candidates per iteration, one per rank; what reaches ``train_agent`` - and therefore PPO's
    RolloutStorage.insert (src/helpers/storage.py:26-34) - is, in order, exactly what the
    reference's own train_agent calls stored for the same samples and rewards
    (controller.json: ppo_rollout, recorded by make_golden.py); the genotype log parses with the
    expressions of src/helpers/num_uq.py:14-16."""
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _util import load_json

    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    log_path = str(tmp_path / "genotypes.out")
    procs = [ctx.Process(target=_ctrl_search_worker, args=(r, world, port, q, log_path)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ctrl = load_json("controller.json")["wacv"]
    rollout, samples = ctrl["ppo_rollout"], ctrl["samples"]
    inserted = results[0]
    assert results[1] == [] and len(inserted) == len(samples) == rollout["step"]
    by_config = {str(s["config"]): s for s in samples}
    for k, (config, reward, entropy, log_prob) in enumerate(inserted):
        # train_agent: action = controller.config2action(config); storage row k = (action, log_prob, reward)
        assert by_config[str(config)]["action"] == rollout["actions"][k], k
        assert abs(reward - rollout["rewards"][k]) < 1e-12
        assert abs(log_prob - rollout["log_probs"][k]) < 1e-6
        assert entropy == by_config[str(config)]["entropy"]
    # the reference's exponential baseline over these rewards (gradient_estimators.py:150-154)
    base = None
    for _, reward, _, _ in inserted:
        base = reward if base is None else 0.95 * base + 0.05 * reward
    assert abs(base - rollout["baseline"]) < 1e-12
    lines = open(log_path, "rb").readlines()
    assert len(lines) == len(samples)
    for k, l in enumerate(lines):
        arch = l.decode("utf-8").strip("\n").split(":")[-1]              # num_uq.py:14
        reward = float(l.decode("utf-8").strip("\n").split(",")[0][7:])  # num_uq.py:15
        epoch = int(l.decode("utf-8").strip("\n").split(":")[2].split(",")[0])  # num_uq.py:16
        assert arch.strip() == str(samples[k]["config"]) and epoch == k
        assert abs(reward - 0.01 * (k + 1)) < 1e-4


def _run_bench(args, env_drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"), **env_set):
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in env_drop}
    env.update(env_set)
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=240)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` started bare spawns two ranks itself (torch.distributed.run, rendezvous on
    127.0.0.1) and reports the number of ranks a REAL all-reduce counted - here on the CPU (gloo, no device):
    the launcher path of the N-GPU benchmark (src/main_search.py:507 is one process with nn.DataParallel)."""
    import json

    out = _run_bench(["--gpus", "2", "--launch-selftest"])
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout  # rank 0 only
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["rccl_ranks"] == 2 and rec["requested"] == 2


def test_bench_refuses_a_world_that_is_not_what_was_asked():
    # a launcher that started fewer ranks than --gpus says must not produce a line that reads as N GPUs
    out = _run_bench(["--gpus", "2", "--launch-selftest"], env_drop=(), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    assert out.returncode != 0 and "WORLD_SIZE=1" in (out.stderr + out.stdout)
    # ... and without devices the real benchmark fails loudly instead of running one rank
    out = _run_bench(["--gpus", "2"])
    if not torch.cuda.is_available():
        assert out.returncode != 0 and "HIP device" in (out.stderr + out.stdout)
