"""N > 1 path on CPU: two gloo ranks render their pixel-tile shards (with the oracle standing in
for the GPU renderer, which is absent here), exchange them with the same pack -> all_gather ->
unpack sequence bench.py uses, and rank 0 must hold the exact single-process frame."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _pack(img, mask_tiles, tile, W, H):
    """Dense tile-major shard buffer [ownedTile][tile*tile][4] in the slot order of
    csrc/pt_kernels.hip slotPixel(): 8x8 pixel blocks inside a tile."""
    tiles_x = (W + tile - 1) // tile
    out = np.zeros((len(mask_tiles), tile * tile, 4), np.float32)
    bpr = tile // 8
    o = np.arange(tile * tile)
    blk, ib = o // 64, o % 64
    lx, ly = (blk % bpr) * 8 + ib % 8, (blk // bpr) * 8 + ib // 8
    for k, t in enumerate(mask_tiles):
        x, y = (t % tiles_x) * tile + lx, (t // tiles_x) * tile + ly
        ok = (x < W) & (y < H)
        out[k, ok] = img[y[ok], x[ok]]
    return out


def _unpack(buf, mask_tiles, tile, W, H, img):
    tiles_x = (W + tile - 1) // tile
    bpr = tile // 8
    o = np.arange(tile * tile)
    blk, ib = o // 64, o % 64
    lx, ly = (blk % bpr) * 8 + ib % 8, (blk // bpr) * 8 + ib // 8
    for k, t in enumerate(mask_tiles):
        x, y = (t % tiles_x) * tile + lx, (t // tiles_x) * tile + ly
        ok = (x < W) & (y < H)
        img[y[ok], x[ok]] = buf[k, ok]


def _worker(rank, world, port, outdir):
    sys.path.insert(0, REPO)
    import __graft_entry__ as graft

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg, orc = graft.load_package(), graft.load_oracle()
    W, H, tile = 100, 70, 32
    s = pkg.Scene("default")
    osc = orc.OracleScene(s.desc)
    acc = np.zeros((H, W, 4), np.float32)
    shard = pkg.TileShard(rank, world, tile)
    for f in range(2):
        osc.render(s.uniform(W, H, bounces=4, total_samples=f), s.lights, W, H, accum=acc, shard=shard, threads=2)
    tiles = [pkg.owned_tiles(W, H, r, world, tile) for r in range(world)]
    n_max = max(len(t) for t in tiles)
    send = torch.zeros(n_max * tile * tile * 4)
    mine = _pack(acc, tiles[rank], tile, W, H)
    send[: mine.size] = torch.from_numpy(mine.reshape(-1))
    recv = torch.zeros(world * send.numel())
    dist.all_gather_into_tensor(recv, send)
    if rank == 0:
        full = np.zeros((H, W, 4), np.float32)
        for r in range(world):
            part = recv[r * send.numel():(r + 1) * send.numel()].numpy()[: len(tiles[r]) * tile * tile * 4]
            _unpack(part.reshape(len(tiles[r]), tile * tile, 4), tiles[r], tile, W, H, full)
        np.save(os.path.join(outdir, "gathered.npy"), full)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_tile_shard_gather(tmp_path, pkg, orc):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "gathered.npy")
    W, H = 100, 70
    s = pkg.Scene("default")
    osc = orc.OracleScene(s.desc)
    ref = np.zeros((H, W, 4), np.float32)
    for f in range(2):
        osc.render(s.uniform(W, H, bounces=4, total_samples=f), s.lights, W, H, accum=ref)
    assert (got.view(np.uint32) == ref.view(np.uint32)).all()
