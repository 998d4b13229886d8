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
        assert dp._flat.numel() == sum(p.numel() for p in reached)
        assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(reached, dp._views))
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
