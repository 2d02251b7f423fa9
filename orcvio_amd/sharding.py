"""Feature / object sharding for the multi-GPU form of the update.

Tracks are independent until the compression step, so they are dealt across ranks balanced by
their projected row count rho_j = 2 M_j - 3; the window, the prior and the gate table are replicated.
"""
from __future__ import annotations

import dataclasses

import numpy as np


def deal_features(obs_ptr: np.ndarray, world: int) -> list:
    """Greedy longest-first dealing of tracks to `world` ranks; returns a list of index arrays
    (ascending within a rank, so every rank keeps the map_server key order)."""
    M = np.diff(obs_ptr)
    rho = np.where(M >= 2, 2 * M - 3, 0)
    order = np.argsort(-rho, kind='stable')
    load = np.zeros(world, dtype=np.int64)
    buckets = [[] for _ in range(world)]
    for j in order:
        r = int(np.argmin(load))
        buckets[r].append(int(j))
        load[r] += int(rho[j])
    return [np.array(sorted(b), dtype=np.int64) for b in buckets]


def shard_window(win, rank: int, world: int):
    """The rank's view of a window: all clones and P, its share of the tracks."""
    if world == 1:
        return win, np.arange(win.F)
    idx = deal_features(win.obs_ptr, world)[rank]
    ptr = [0]
    sel = []
    for j in idx:
        sel += list(range(int(win.obs_ptr[j]), int(win.obs_ptr[j + 1])))
        ptr.append(len(sel))
    sel = np.asarray(sel, dtype=np.int64)
    sub = dataclasses.replace(win, p_w=win.p_w[idx].copy(), obs_ptr=np.asarray(ptr, dtype=np.int32),
                              obs_clone=win.obs_clone[sel].copy(), obs_z=win.obs_z[sel].copy(),
                              obs_zvel=win.obs_zvel[sel].copy())
    return sub, idx


def sum_blocks(blocks):
    """Rank-ordered sum of the gathered compressed blocks (same order on every rank -> identical bits)."""
    out = blocks[0].copy() if hasattr(blocks[0], 'copy') else blocks[0].clone()
    for b in blocks[1:]:
        out += b
    return out
