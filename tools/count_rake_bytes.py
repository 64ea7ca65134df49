#!/usr/bin/env python
"""Algorithmic bytes of ONE soil_accumulate call (csrc/graph.hip: k_donors4 + k_rake_compress rounds) on
BASELINE config 3's kind of graph — counted from the kernel's own state machine, on the CPU, with numpy.

"Algorithmic" = every distinct 4-byte word the kernels of the call read or write, counted once per kernel
launch that touches it, and nothing else: no sector granularity, no re-reads.  That is the figure bench.py
--config c3 prices the call with (`roofline.algorithmic_bytes_per_cell`); the counters of
tools/pmc_rake.sh say what the memory system really moved for it.

  k_donors4   reads graph (3 rows of it, each word once) 4 and source 4; writes count 4, value 4 and one
              donor slot per edge 4 * edges / cell                                        (decay-free call)
  a round     every cell reads its count 4 (a LISTED round, from --list-from on: only the cells on its lists, and
              their list entries 4; a cell that has to be looked at again costs a list entry 4 more).  A cell with count >= 0 reads its value 4 and its `count` donor
              slots; per donor it gathers the donor's count 4, then (donor final or single-donor) the donor's
              value 4 and (single-donor) the donor's slot 0 4; it writes value 4 and, if it was final already,
              count in both buffers 8, else count 4 and its remaining slots.
              A round whose predecessor left nothing pending reads one flag and returns: 0 bytes.

Usage: python tools/count_rake_bytes.py [--size 4096] [--k 0] [--out profiles/r06_accumulate/algorithmic_bytes.json]
(CPU only: the DEM, the pit fill and random_weighted come from the oracle; ~2 min at 4096^2.)
"""
import argparse
import json
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as o  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=4096)
ap.add_argument("--k", type=int, default=0, help="the realisation (random_weighted's offset)")
ap.add_argument("--list-from", type=int, default=2, help="the first listed round (csrc/graph.hip: SOIL_RAKE_LIST_FROM); 0: dense rounds throughout")
ap.add_argument("--out", default="")
args = ap.parse_args()
S = args.size
o.set_threads(8)
dem = o.noise(S, S, seed=3.0, ext=(float(S), float(S))) * np.float32(100.0)
dem = o.fill_depressions(dem, 1)
graph = o.random_weighted(dem, 1, 0, args.k, 10.0).reshape(-1).astype(np.int64)
elem = S * S
has = graph >= 0
edges = int(has.sum())
# donors of every cell, as a CSR over receivers (slot order does not change the byte count)
order = np.argsort(graph[has], kind="stable")
src = np.nonzero(has)[0][order]
dst = graph[has][order]
count = np.bincount(dst, minlength=elem).astype(np.int64)
start = np.concatenate(([0], np.cumsum(count)))[:-1]
Kmax = int(count.max())
slots = np.full((Kmax, elem), -1, np.int64)
for k in range(Kmax):
    sel = count > k
    slots[k, sel] = src[start[sel] + k]
setup = 4 + 4 + 4 + 4 + 4.0 * edges / elem
print("graph: %d cells, %d edges (%.3f per cell), max in-degree %d; k_donors4 %.2f B/cell" % (elem, edges, edges / elem, Kmax, setup))
rounds = 2 * (math.ceil(math.log2(elem) / 2) + 1)
cnt = count.copy()          # state in the `in` buffer: > 0 pending, 0 final (other buffer stale), -1 final in both
per_round = []
work_left = True
for r in range(rounds):
    if not work_left:
        per_round.append(0.0)
        continue
    act = cnt >= 0
    n_act = int(act.sum())
    listed = args.list_from >= 1 and r >= args.list_from
    # a dense round: every cell reads its count; a listed one: its cells' list entries and counts
    b = 8.0 * n_act if listed else 4.0 * elem
    b += 4.0 * n_act + 4.0 * int(cnt[act].sum())          # value + its slots
    new_cnt = cnt.copy()
    new_slots = slots.copy()
    keep = np.zeros(elem, np.int64)
    out_slots = np.full_like(slots, -1)
    for k in range(Kmax):
        sel = act & (cnt > k)
        d = slots[k, sel]
        dc = cnt[d]
        b += 4.0 * d.size                                   # the donor's count
        fin = dc <= 0
        one = dc == 1
        b += 4.0 * int(fin.sum()) + 8.0 * int(one.sum())    # its value (+ its slot 0)
        nd = d.copy()
        nd[one] = slots[0, d[one]]
        kept = ~fin
        idx = np.nonzero(sel)[0][kept]
        out_slots[keep[idx], idx] = nd[kept]
        keep[idx] += 1
    was_final = act & (cnt == 0)
    b += 4.0 * n_act                                        # value out
    b += 8.0 * int(was_final.sum())                         # count = -1 in both buffers
    pend = act & ~was_final
    b += 4.0 * int(pend.sum()) + 4.0 * int(keep[pend].sum())
    if args.list_from >= 1 and r + 1 >= args.list_from:
        b += 4.0 * int(pend.sum())                          # the entry on the next round's list
    new_cnt[was_final] = -1
    new_cnt[pend] = keep[pend]
    slots = np.where(pend[None, :], out_slots, slots)
    cnt = new_cnt
    work_left = bool(pend.any())
    per_round.append(b / elem)
    print("round %2d: %8.2f B/cell   active %5.1f %%  pending after %5.1f %%" % (r, b / elem, 100.0 * n_act / elem, 100.0 * pend.sum() / elem))
total = setup + sum(per_round)
print("one accumulate call: %.2f B/cell (k_donors4 %.2f + rounds %.2f over %d live rounds of %d)" % (
    total, setup, sum(per_round), sum(1 for v in per_round if v > 0), rounds))
if args.out:
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump({"grid": [S, S], "realisation": args.k, "edges_per_cell": edges / elem,
               "bytes_per_cell_k_donors4": setup, "bytes_per_cell_rounds": per_round,
               "bytes_per_cell_accumulate": total, "list_from": args.list_from,
               "how": "tools/count_rake_bytes.py: distinct words read or written per kernel launch by k_donors4 and the "
                      "k_rake_compress rounds of one decay-free accumulate on realisation %d of the %dx%d D8 "
                      "random_weighted(T=10) graph, counted from the kernel's state machine with numpy" % (args.k, S, S)},
              open(args.out, "w"), indent=1)
