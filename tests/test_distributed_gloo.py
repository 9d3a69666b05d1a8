"""CPU, world_size 2, gloo: the data-parallel gradient exchange used by bench.py for N > 1
(face_mask_inpaint_amd/distributed.py).  Two ranks with different data must end up with identical parameters,
equal to one process that saw both shards (mean-reduced losses), including a parameter that never receives a
gradient (Auto_Attn.alpha-like) and bucket boundaries that split the gradient list."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model():
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 3))
    m.unused = torch.nn.Parameter(torch.zeros(1))  # never receives a gradient
    return m


def _data(rank):
    g = torch.Generator().manual_seed(100 + rank)
    return torch.randn(5, 6, generator=g), torch.randn(5, 3, generator=g)


def _worker(rank, world, port, out, exchange="allreduce"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from face_mask_inpaint_amd import distributed as fd

    m = _model()
    if rank == 1:  # start from different weights: broadcast must fix that
        with torch.no_grad():
            for p in m.parameters():
                p.add_(1.0)
    fd.broadcast_parameters([m])
    opt = fd.DataParallelOptimizer(torch.optim.Adam(m.parameters(), lr=1e-2), bucket_bytes=256, exchange=exchange)  # forces several buckets
    x, y = _data(rank)
    for _ in range(3):
        opt.zero_grad()
        torch.nn.functional.mse_loss(m(x), y).backward()
        opt.step()
    assert fd.allreduce_gradients(m.parameters(), bucket_bytes=256) >= 2
    torch.save({k: v.clone() for k, v in m.state_dict().items()}, out + f".{rank}")
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["allreduce", "direct", "rs_ag"])
def test_two_rank_gradient_exchange(tmp_path, exchange):
    """exchange: one all-reduce per bucket, or the point-to-point reduce-scatter + all-gather (all-to-all of the shards, local sum,
    all-gather; "rs_ag" takes that form on gloo, which has no reduce-scatter) with bucket lengths that are not multiples of the world size"""
    port, out = _free_port(), str(tmp_path / "sd")
    mp.spawn(_worker, args=(2, port, out, exchange), nprocs=2, join=True)
    a, b = torch.load(out + ".0"), torch.load(out + ".1")
    for k in a:
        assert torch.equal(a[k], b[k]), k
    # single process on the union of the shards: mean of the two per-shard mean losses
    m = _model()
    opt = torch.optim.Adam(m.parameters(), lr=1e-2)
    (x0, y0), (x1, y1) = _data(0), _data(1)
    for _ in range(3):
        opt.zero_grad()
        (0.5 * (torch.nn.functional.mse_loss(m(x0), y0) + torch.nn.functional.mse_loss(m(x1), y1))).backward()
        opt.step()
    for k, v in m.state_dict().items():
        torch.testing.assert_close(a[k], v, rtol=1e-5, atol=1e-6)
    assert float(a["unused"]) == 0.0


def _gan_worker(rank, world, port, out):
    """two wrapped optimisers in the early-discriminator order GANOptimizer uses when data-parallel: D backward, D launch()
    (non-blocking), G backward with hook-driven buckets, G step, D step"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from face_mask_inpaint_amd import distributed as fd

    G, D = _model(), _model()
    fd.broadcast_parameters([G, D])
    og = fd.DataParallelOptimizer(torch.optim.Adam(G.parameters(), lr=1e-2), bucket_bytes=200)
    od = fd.DataParallelOptimizer(torch.optim.Adam(D.parameters(), lr=1e-2), bucket_bytes=1 << 20)  # never fills: needs launch()
    x, y = _data(rank)
    for _ in range(2):
        gen = G(x)
        for p in D.parameters():  # the generator pass must not produce D gradients (GANOptimizer does the same)
            p.requires_grad_(False)
        g_loss = torch.nn.functional.mse_loss(gen, y) + 0.1 * D(torch.cat([gen, gen], 1)).square().mean()
        for p in D.parameters():
            p.requires_grad_(True)
        d_loss = D(torch.cat([gen.detach(), y], 1)).square().mean()
        od.zero_grad()
        d_loss.backward()
        od.launch()
        assert len(od._inflight) == 1 and not od._open
        og.zero_grad()
        g_loss.backward()
        og.step()
        od.step()
    assert og.collectives >= 4 and od.collectives == 2
    torch.save({"G": G.state_dict(), "D": D.state_dict()}, out + f".{rank}")
    dist.destroy_process_group()


def test_two_rank_early_discriminator_schedule(tmp_path):
    port, out = _free_port(), str(tmp_path / "gan")
    mp.spawn(_gan_worker, args=(2, port, out), nprocs=2, join=True)
    a, b = torch.load(out + ".0"), torch.load(out + ".1")
    for net in ("G", "D"):
        for k in a[net]:
            assert torch.equal(a[net][k], b[net][k]), (net, k)
    # reference order in one process on the union of the shards: G backward, G step, then D loss / backward / step
    G, D = _model(), _model()
    og, od = torch.optim.Adam(G.parameters(), lr=1e-2), torch.optim.Adam(D.parameters(), lr=1e-2)
    data = [_data(0), _data(1)]
    for _ in range(2):
        gens = [G(x) for x, _ in data]
        g_loss = sum(torch.nn.functional.mse_loss(g, y) + 0.1 * D(torch.cat([g, g], 1)).square().mean() for g, (_, y) in zip(gens, data)) / 2
        og.zero_grad()
        g_loss.backward()
        og.step()
        d_loss = sum(D(torch.cat([g.detach(), y], 1)).square().mean() for g, (_, y) in zip(gens, data)) / 2
        od.zero_grad()
        d_loss.backward()
        od.step()
    for net, m in (("G", G), ("D", D)):
        for k, v in m.state_dict().items():
            torch.testing.assert_close(a[net][k], v, rtol=1e-5, atol=1e-6, msg=lambda s, k=k: f"{net}.{k}: {s}")


class _StubVGG(torch.nn.Module):
    """CPU stand-in for VGGLoss.forward_multi (the HIP feature extractor cannot run here): smooth per-term losses of the same
    arguments, so that GANOptimizer.__call__ -- the schedule under test -- runs unmodified"""

    def forward_multi(self, jobs):
        return [((a - b).square().mean() if kind == "perceptual" else (a.mean(dim=(2, 3)) - b.mean(dim=(2, 3))).abs().mean()) for a, b, kind in jobs]


def _cpu_gan_optimizer(optD, optG):
    from face_mask_inpaint_amd.modules.loss import GANOptimizer

    class CpuGANOptimizer(GANOptimizer):  # only the HIP-backed loss terms are replaced; __call__ is the product's
        @staticmethod
        def _masked(img, mask, invert):
            m = mask.unsqueeze(1)
            return img * ((1 - m) if invert else m)

        def discriminator_loss(self, netD, real, fake):
            return 0.5 * ((netD(real) - 1).square().mean() + netD(fake.detach()).square().mean())

        def generator_loss(self, netD, real, fake, freeze=True):
            return (netD(fake) - 1).square().mean() * self.lambda_g + (fake - real).abs().mean()

    gopt = CpuGANOptimizer(optD, optG, vgg_width_div=16)
    gopt.vgg_loss = _StubVGG()
    return gopt


def _conv_nets():
    torch.manual_seed(0)
    G = torch.nn.Sequential(torch.nn.Conv2d(6, 8, 3, padding=1), torch.nn.Tanh(), torch.nn.Conv2d(8, 3, 3, padding=1))
    D = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.LeakyReLU(0.1), torch.nn.Conv2d(8, 8, 3, padding=1), torch.nn.LeakyReLU(0.1),
                            torch.nn.Conv2d(8, 1, 3))
    D.alpha = torch.nn.Parameter(torch.zeros(1))  # Auto_Attn.alpha-like: never receives a gradient
    return G, D


def _gan_batch(rank):
    g = torch.Generator().manual_seed(200 + rank)
    src, ref, gt = (torch.rand(3, 3, 12, 12, generator=g) for _ in range(3))
    return src, ref, gt, (torch.rand(3, 12, 12, generator=g) < 0.4).float()


def _real_gan_worker(rank, world, port, out):
    """the product's GANOptimizer.__call__ (early-discriminator branch, chosen automatically because the optimisers are
    DataParallelOptimizer) with a D bucket size that makes the hooks launch buckets in the MIDDLE of the D backward and leaves a
    remainder for launch()"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from face_mask_inpaint_amd import distributed as fd

    G, D = _conv_nets()
    fd.broadcast_parameters([G, D])
    og = fd.DataParallelOptimizer(torch.optim.Adam(G.parameters(), lr=1e-2), bucket_bytes=1024)
    od = fd.DataParallelOptimizer(torch.optim.Adam(D.parameters(), lr=1e-2), bucket_bytes=1500)  # 3 conv layers: 896 + 2336 + 292 bytes
    gopt = _cpu_gan_optimizer(od, og)
    assert gopt.early_d is None  # automatic
    src, ref, gt, mask = _gan_batch(rank)
    sent_mid_backward = []
    orig_launch = od.launch

    def launch():
        sent_mid_backward.append(len(od._inflight))  # buckets already in flight when launch() runs (the hooks call it for full buckets too)
        return orig_launch()

    od.launch = launch
    losses = []
    for _ in range(2):
        gen = G(torch.cat([src, ref], 1))
        losses.append([float(v) for v in gopt(D, src, gt, ref, gen, mask)])
    # per step: two hook-driven launches inside the D backward (0, then 1 bucket in flight), then GANOptimizer's explicit launch()
    # of the remainder with 2 in flight -- a D bucket boundary fell inside the D backward
    assert sent_mid_backward[:3] == [0, 1, 2], sent_mid_backward
    assert od.collectives >= 4 and og.collectives >= 2
    # a backward without a step (validation / skipped step) must not poison the next step: zero_grad drops its buckets
    gen = G(torch.cat([src, ref], 1))
    (D(gen)).mean().backward()
    assert od._inflight or od._open
    od.zero_grad(set_to_none=True)
    og.zero_grad(set_to_none=True)
    assert not od._inflight and not od._open
    gen = G(torch.cat([src, ref], 1))
    losses.append([float(v) for v in gopt(D, src, gt, ref, gen, mask)])
    torch.save({"G": G.state_dict(), "D": D.state_dict(), "losses": losses}, out + f".{rank}")
    dist.destroy_process_group()


def test_real_gan_optimizer_early_discriminator_two_ranks(tmp_path):
    port, out = _free_port(), str(tmp_path / "rgan")
    mp.spawn(_real_gan_worker, args=(2, port, out), nprocs=2, join=True)
    a, b = torch.load(out + ".0"), torch.load(out + ".1")
    for net in ("G", "D"):
        for k in a[net]:
            assert torch.equal(a[net][k], b[net][k]), (net, k)
    # one process, reference order (early_d off), on the union of the two shards (mean over the ranks of every loss)
    G, D = _conv_nets()
    og, od = torch.optim.Adam(G.parameters(), lr=1e-2), torch.optim.Adam(D.parameters(), lr=1e-2)

    class Both:  # evaluates the per-rank mean losses of both shards: the sum of gradients / 2 equals the all-reduced average
        pass

    gopt = _cpu_gan_optimizer(od, og)
    gopt.early_d = False
    batches = [_gan_batch(0), _gan_batch(1)]
    for it in range(3):
        # G step
        gens = [G(torch.cat([s, r], 1)) for s, r, _, _ in batches]
        g_total = 0
        for gen, (src, ref, gt, mask) in zip(gens, batches):
            for p in D.parameters():
                p.requires_grad_(False)
            gl = gopt.generator_loss(D, gt, gen, freeze=False)
            for p in D.parameters():
                p.requires_grad_(True)
            perc, sty, cx = gopt.vgg_loss.forward_multi([(gen, gt, "perceptual"), (gopt._masked(gen, mask, True), src, "style"),
                                                         (gopt._masked(gen, mask, False), gopt._masked(ref, mask, False), "contextual")])
            g_total = g_total + (gl + perc * gopt.lambda_perc + sty * gopt.lambda_style + cx * gopt.lambda_cx) / 2
        og.zero_grad()
        g_total.backward()
        og.step()
        d_total = sum(gopt.discriminator_loss(D, gt, gen) for gen, (_, _, gt, _) in zip(gens, batches)) / 2
        od.zero_grad()
        d_total.backward()
        od.step()
    for net, m in (("G", G), ("D", D)):
        for k, v in m.state_dict().items():
            torch.testing.assert_close(a[net][k], v, rtol=1e-5, atol=1e-6, msg=lambda s, k=k: f"{net}.{k}: {s}")
    assert float(a["D"]["alpha"]) == 0.0
