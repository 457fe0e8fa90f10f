"""world_size-2 gloo test of the sharding path (no GPU): each rank fills the slab of its
tiles, the slabs are all_gathered and de-tiled, and the result must be bit-identical to the
unsharded image.  Pixel values come from the oracle (test infrastructure) because no GPU is
present here; what is under test is volxel_amd.tiles / volxel_amd.dist."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, w, h, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from volxel_amd import tiles
    from volxel_amd.dist import gather_image
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = np.load(os.path.join(out_dir, "full.npy"))
    for tag, perm in (("", None), ("_bal", np.load(os.path.join(out_dir, "perm.npy")))):
        px, py = tiles.slab_pixel_coords(w, h, rank, world, perm)
        slab = np.zeros((px.size, 4), dtype=np.float32)
        ok = px >= 0
        slab[ok] = full[py[ok], px[ok]]
        img = gather_image(torch.from_numpy(slab.reshape(-1)), w, h, perm=perm)
        np.save(os.path.join(out_dir, f"img{rank}{tag}.npy"), img.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("w,h", [(200, 136), (128, 64)])
def test_two_rank_gather_is_bit_identical(oracle, tmp_path, w, h):
    import torch.multiprocessing as mp
    from tests.common import make_scene
    from volxel_amd import synth, default_transfer_function
    vox, sp = synth.sphere(32)
    g = oracle.BrickGrid(vox, sp)
    tf, L = default_transfer_function()
    s, cam, vol, ds, p = make_scene(g, w, h, "dvr")
    full, _ = oracle.render(p, g, tf, L)
    np.save(tmp_path / "full.npy", full)
    from volxel_amd import tiles
    nt = tiles.tile_counts(w, h, 2)[2]
    costs = np.random.default_rng(4).integers(0, 1000, size=nt)      # any costs: the order is a permutation
    np.save(tmp_path / "perm.npy", tiles.balanced_order(costs, 2))
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, w, h, str(tmp_path)), nprocs=2, join=True)
    for rank in range(2):
        assert np.array_equal(np.load(tmp_path / f"img{rank}.npy"), full)
        assert np.array_equal(np.load(tmp_path / f"img{rank}_bal.npy"), full)    # balanced dealing order


def test_tile_partition_covers_every_pixel_once():
    from volxel_amd import tiles
    for (w, h, n) in [(1920, 1080, 1), (1920, 1080, 8), (3840, 2160, 8), (100, 70, 3), (64, 64, 2)]:
        seen = np.zeros((h, w), dtype=np.int32)
        tx, ty, nt, tps = tiles.tile_counts(w, h, n)
        for r in range(n):
            px, py = tiles.slab_pixel_coords(w, h, r, n)
            assert px.size == tps * 4096
            ok = px >= 0
            np.add.at(seen, (py[ok], px[ok]), 1)
        assert (seen == 1).all()
        # a balanced dealing order is a permutation: the same holds, and the cost sums level out
        costs = np.random.default_rng(n).gamma(2.0, 100.0, size=nt).astype(np.int64)
        perm = tiles.balanced_order(costs, n)
        assert sorted(perm.tolist()) == list(range(nt))
        seen[:] = 0
        for r in range(n):
            px, py = tiles.slab_pixel_coords(w, h, r, n, perm)
            ok = px >= 0
            np.add.at(seen, (py[ok], px[ok]), 1)
        assert (seen == 1).all()
        if n > 1 and nt >= 8 * n:
            dealt = [costs[perm[r::n]].sum() for r in range(n)]
            plain = [costs[r::n].sum() for r in range(n)]
            assert max(dealt) / np.mean(dealt) <= max(plain) / np.mean(plain) + 1e-9
            assert max(dealt) / np.mean(dealt) < 1.02
