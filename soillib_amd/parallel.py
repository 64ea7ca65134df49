"""Row-slab sharding of the erosion step across the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" == RCCL over xGMI).
Rank r owns global rows [r*S, (r+1)*S) of a (world*S, W) grid and holds G ghost
rows on each interior side, G = soil_ghost_rows(param) = ceil(sqrt(2)*maxage)+2:
one particle step moves at most sqrt(2) cells (erosion_map.cu:61-76), so no
trajectory born in the owned rows can leave the slab.  That makes the sharded
step EXACT (same trajectories, same deposits as the single-GPU run; only the
fp32 summation order of the flux differs) with nearest-neighbour traffic only:

  per step                         exchanged with each neighbour
  1 fluvial particles              -
  2 debris particles               overlapped: flux halo-accumulate of the fluvial planes
                                   (G rows x 4 floats -> added by the owner)
  3 flux halo of the debris planes G rows x 3 floats (exposed)
  4 cell phase, bands next to the  -
    neighbours
  5 cell phase, interior rows      overlapped: field halo — G rows of layers, velocity,
                                   waterHeight, debrisVelocity -> neighbour's ghost rows

How much of the G rows really travels is decided by measurement, not by the bound: after each
particle launch a rank looks how deep into its ghost rows the deposits got (`ghost_extent`; at
the acceptance parameters ~190 of the 365 rows for water, none for debris) and ships exactly
those rows of flux; the field halo is refreshed as deep as the walks of the last steps reached,
plus a margin, and a launch whose deposits get within a row of the refreshed depth is REPEATED
after the missing rows have been fetched (they are still unchanged at the neighbour's: the cell
phase has not run yet), so the step stays exact whatever the prediction was.  The reach numbers
of all ranks travel in two small all-reduces per step.

Every rank replays all world*N particle streams (two Philox draws each) and
traces the ones whose spawn row it owns (soil_particles_*_slab), so the set of
trajectories is identical to a single-GPU run of the global grid.  Even the
reference's NaN walkers, whose one deposit belongs
to global cell (0,0), are reproduced: a rank that does not hold global row 0
parks those deposits in an 8-float buffer that is all-reduced to the owner.

The runner is written against a small `ops` interface so that the exchange and
partition logic can be tested on CPU (gloo) with the oracle as the compute
back-end (tests/test_parallel_gloo.py); the product back-end is HipOps — HIP
kernels through the C ABI, torch only for allocation and communication.
"""
import ctypes as C
import os

FIELD_PLANES = ("layers", "velocity", "waterHeight", "debrisVelocity")
FLUX_FLUVIAL = ("waterFlux", "massFlux", "velocityFlux")      # final after the fluvial launch
FLUX_DEBRIS = ("debrisFlux", "debrisVelocityFlux")            # final after the debris launch
FLUX_PLANES = FLUX_FLUVIAL + FLUX_DEBRIS
PLANE_CHANNELS = {
    "layers": 2, "layers_next": 2, "height": 1, "uplift": 1, "rainfall": 1, "waterHeight": 1,
    "waterFlux": 1, "mass": 1, "massFlux": 1, "velocity": 2, "velocityFlux": 2, "debris": 1,
    "debrisFlux": 1, "debrisVelocity": 2, "debrisVelocityFlux": 2,
}


def slab_layout(rank, world, S, G):
    """Rows a rank holds: (x0, rows, r0, r1) — global row of local row 0, local
    row count, owned local row range."""
    H = world * S
    o0, o1 = rank * S, (rank + 1) * S
    x0 = max(0, o0 - G)
    x1 = min(H, o1 + G)
    return x0, x1 - x0, o0 - x0, o1 - x0


class HipOps:
    """Product back-end: torch CUDA tensors for storage, HIP kernels via the C ABI."""

    def __init__(self, local_rank):
        import torch
        from . import _abi
        self.torch, self.abi, self.lib = torch, _abi, _abi.lib()
        torch.cuda.set_device(local_rank)
        _abi.check(self.lib.soil_set_device(local_rank))
        self.device = torch.device("cuda", local_rank)
        self.main = torch.cuda.current_stream()
        self.comm = torch.cuda.Stream()

    def alloc(self, shape, kind="f32"):
        t = self.torch
        if kind == "rng":
            return t.zeros((shape[0], 2), dtype=t.int64, device=self.device)
        return t.zeros(tuple(shape), dtype=t.float32, device=self.device)

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream().cuda_stream)

    @staticmethod
    def _p(t):
        return C.c_void_p(t.data_ptr())

    def seed(self, rng, seed, offset):
        self.abi.check(self.lib.soil_rng_seed(self._p(rng), rng.shape[0], seed, offset,
                                              self._stream()))

    def fill(self, t, value):
        self.abi.check(self.lib.soil_set_f32(self._p(t), float(value), t.numel(), self._stream()))

    def zero(self, t):
        self.fill(t, 0.0)

    def add(self, dst, src):
        self.abi.check(self.lib.soil_add_f32(self._p(dst), self._p(src), dst.numel(),
                                             self._stream()))

    def copy(self, dst, src):
        self.abi.check(self.lib.soil_memcpy_d2d(self._p(dst), self._p(src),
                                                dst.numel() * dst.element_size(), self._stream()))

    def noise_rows(self, out, H, W, x0, seed):
        """Bedrock rows [x0, x0+rows) of the global soil.noise heightmap."""
        from . import _abi
        p = _abi.NoiseParam()
        self.lib.soil_noise_param_default(C.byref(p))
        p.seed = seed
        p.ext[0], p.ext[1] = float(H), float(W)
        # noise is a pure function of the global cell index: generate the whole
        # columns x rows window by offsetting the row origin
        self.abi.check(self.lib.soil_noise_window(self._p(out), out.shape[0], W, x0, C.byref(p),
                                                  self._stream()))

    def layers_from_bedrock(self, layers, bed):
        self.abi.check(self.lib.soil_layers_from_planes(self._p(layers), self._p(bed), None,
                                                        bed.numel(), self._stream()))

    def particles_fluvial(self, P, rng, N, dom, scale, param, remote0):
        self.abi.check(self.lib.soil_particles_fluvial_slab(
            self._p(P["waterFlux"]), self._p(P["massFlux"]), self._p(P["velocityFlux"]), None,
            self._p(rng), N, self._p(P["layers"]), self._p(P["rainfall"]),
            self._p(P["waterHeight"]), self._p(P["velocity"]), None, self._p(remote0),
            C.byref(dom), self.abi.vec(scale, 3), param._ref(), self._stream()))

    def particles_debris(self, P, rng, N, dom, scale, param, remote0):
        self.abi.check(self.lib.soil_particles_debris_slab(
            self._p(P["debrisFlux"]), self._p(P["debrisVelocityFlux"]), None, self._p(rng), N,
            self._p(P["layers"]), self._p(P["debrisVelocity"]), None, self._p(remote0),
            C.byref(dom), self.abi.vec(scale, 3), param._ref(), self._stream()))

    def particles_pair(self, P, rng, rng_debris, N, dom, scale, param, remote0):
        """Both launches overlapped (soil_particles_pair_slab)."""
        planes = self.abi.ErosionPlanes()
        for name in self.abi._PLANES:
            setattr(planes, name, P[name].data_ptr())
        self.abi.check(self.lib.soil_particles_pair_slab(
            C.byref(planes), self._p(rng), self._p(rng_debris), N, self._p(remote0), C.byref(dom),
            self.abi.vec(scale, 3), param._ref(), self._stream()))

    def ghost_extent(self, planes, r0, r1):
        """(rows above, rows below) the owned local rows [r0, r1) that hold a deposit in any of
        `planes` — soil_ghost_extent; synchronises the current stream (two ints come back)."""
        t = self.torch
        if not hasattr(self, "_extent"):
            self._extent = t.zeros(2, dtype=t.int32, device=self.device)
        self._extent.zero_()
        for p in planes:
            self.abi.check(self.lib.soil_ghost_extent(
                self._p(self._extent), self._p(p), p.shape[0], p.numel() // p.shape[0], r0, r1,
                self._stream()))
        up, down = self._extent.tolist()
        return int(up), int(down)

    def add_cell0(self, P, remote0):
        """Global cell (0,0) += the all-reduced deposits of the other ranks' NaN walkers."""
        for plane, lo, n in (("waterFlux", 0, 1), ("massFlux", 1, 1), ("velocityFlux", 2, 2),
                             ("debrisFlux", 4, 1), ("debrisVelocityFlux", 5, 2)):
            self.abi.check(self.lib.soil_add_f32(self._p(P[plane]),
                                                 C.c_void_p(remote0.data_ptr() + 4 * lo), n,
                                                 self._stream()))

    def cells(self, P, dom, r0, r1, scale, param):
        if r1 <= r0:
            return
        planes = self.abi.ErosionPlanes()
        for name in self.abi._PLANES:
            setattr(planes, name, P[name].data_ptr())
        d = self.abi.Domain(dom.H, dom.W, dom.x0, dom.rows, r0, r1)
        self.abi.check(self.lib.soil_erode_cells_fused(C.byref(planes), C.byref(d),
                                                       self.abi.vec(scale, 3), param._ref(),
                                                       self._stream()))

    # -- stream plumbing for overlap ---------------------------------------
    def fork_comm(self):
        """Make the communication stream wait for everything queued so far."""
        self.comm.wait_stream(self.torch.cuda.current_stream())
        return self.torch.cuda.stream(self.comm)

    def join_comm(self):
        self.torch.cuda.current_stream().wait_stream(self.comm)

    def sync(self):
        self.torch.cuda.synchronize()


class SlabRunner:
    """The sharded erosion model; `step()` advances the global grid by one step."""

    def __init__(self, rows_per_rank, W, param, particles_div=8, seed=0, ops=None, scale=None,
                 noise_seed=3.0, init=True, comm=None, rank=None, world=None, noise_rows=None):
        """`comm` is a torch.distributed-like module (P2POp, isend, irecv,
        batch_isend_irecv, all_reduce, barrier); the default is torch.distributed
        itself.  Tests inject an in-process stand-in to drive several slabs on one GPU.
        `noise_rows`: the row extent the initial noise heightmap is normalised by (default:
        the global height, i.e. the same landscape stretched over more rows as the world
        grows; weak-scaling runs pass `rows_per_rank` to keep the terrain statistics per cell)."""
        if comm is None:
            import torch.distributed as dist
        else:
            dist = comm
        self.dist = dist
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        # SOIL_DEVICE / SOIL_DIST_BACKEND: which GPU and which torch.distributed backend, when
        # they are not LOCAL_RANK and RCCL (several ranks sharing one GPU over gloo: tests)
        local_rank = int(os.environ.get("SOIL_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        if ops is None:
            ops = HipOps(local_rank)
        self.ops = ops
        if comm is None and not dist.is_initialized():
            backend = os.environ.get("SOIL_DIST_BACKEND") or (
                "nccl" if isinstance(ops, HipOps) else "gloo")
            kw = {}
            if backend == "nccl":
                kw["device_id"] = ops.device
            dist.init_process_group(backend=backend, **kw)
        # RCCL orders its transfers with the stream they are issued on; gloo moving device
        # tensors (tests: several ranks on one GPU) does not, so the host waits for the
        # device before every exchange there
        self._host_ordered = (comm is None and isinstance(ops, HipOps)
                              and dist.get_backend() != "nccl")
        self.S, self.W, self.param = int(rows_per_rank), int(W), param
        self.H = self.world * self.S
        self.scale = list(scale) if scale is not None else [20.0 / self.H, 20.0 / self.W, 4.0]
        self.G = int(ops.ghost_rows(param)) if hasattr(ops, "ghost_rows") else self._ghost(param)
        if self.world > 1 and self.G > self.S:
            raise ValueError("ghost depth %d exceeds the %d rows a neighbour owns" %
                             (self.G, self.S))
        self.x0, self.rows, self.r0, self.r1 = slab_layout(self.rank, self.world, self.S, self.G)
        self.N = self.H * self.W // int(particles_div)   # particles of the GLOBAL grid
        self.seed = int(seed)
        self.step_index = 0
        self.dom = self._domain(self.r0, self.r1)
        self.P = {name: ops.alloc((self.rows, self.W, ch) if ch > 1 else (self.rows, self.W))
                  for name, ch in PLANE_CHANNELS.items()}
        self.rng = ops.alloc((self.N,), "rng")
        self.rng_debris = ops.alloc((self.N,), "rng") if hasattr(ops, "particles_pair") else None
        self.serial_particles = os.environ.get("SOIL_STEP_PAIR") != "1"   # overlap is opt-in
        self.remote0 = ops.alloc((8,))
        self.up = self.rank - 1 if self.rank > 0 else None
        self.down = self.rank + 1 if self.rank < self.world - 1 else None
        # staging buffers for the flux halo-accumulate (one per plane and side)
        self.gu, self.gd = self.r0, self.rows - self.r1      # ghost rows above / below
        # measured-reach trimming of the halos (module docstring); SOIL_HALO_FULL=1: always G rows
        self.trim = (self.world > 1 and hasattr(ops, "ghost_extent")
                     and os.environ.get("SOIL_HALO_FULL") != "1")
        self.fresh_up, self.fresh_down = self.gu, self.gd   # ghost rows whose fields are up to date
        self.reach_hist = []     # max reach over all ranks, last steps (the same on every rank)
        self.fallbacks = 0       # launches repeated because the refreshed depth was too small
        self.halo_rows = {"flux": 0, "field": 0, "full": 0}  # rows shipped so far vs the bound
        self._ints = None
        # (ghost rows above, below) whose fields every rank holds up to date: all of them at first
        self._fresh_all = [(slab_layout(r, self.world, self.S, self.G)[2],
                            slab_layout(r, self.world, self.S, self.G)[1] -
                            slab_layout(r, self.world, self.S, self.G)[3]) for r in range(self.world)]
        self.stage = {}
        for name in FLUX_PLANES:
            ch = PLANE_CHANNELS[name]
            tail = (self.W, ch) if ch > 1 else (self.W,)
            self.stage[name] = (ops.alloc((self._peer_ghost(self.up),) + tail) if self.up is not None else None,
                                ops.alloc((self._peer_ghost(self.down),) + tail) if self.down is not None else None)
        if init:
            bed = ops.alloc((self.rows, self.W))
            ops.noise_rows(bed, self.H if noise_rows is None else int(noise_rows), self.W, self.x0,
                           noise_seed)
            ops.layers_from_bedrock(self.P["layers"], bed)
            self.fill(self.P["rainfall"], 1.0)

    # -- helpers -------------------------------------------------------------
    def _ghost(self, param):
        from . import _abi
        return _abi.lib().soil_ghost_rows(param._ref())

    def _domain(self, r0, r1):
        from . import _abi
        return _abi.Domain(self.H, self.W, self.x0, self.rows, r0, r1)

    def _peer_ghost(self, peer):
        """Ghost rows the neighbour `peer` holds on the side facing this rank."""
        if peer is None:
            return 0
        x0, rows, r0, r1 = slab_layout(peer, self.world, self.S, self.G)
        return r0 if peer > self.rank else rows - r1

    def fill(self, t, value):
        self.ops.fill(t, value)

    # -- halo exchanges --------------------------------------------------------
    def _exchange(self, sends, recvs):
        """sends/recvs: lists of (tensor_view, peer).  One batched group per call."""
        dist = self.dist
        ops_ = [dist.P2POp(dist.isend, t, peer) for t, peer in sends] + \
               [dist.P2POp(dist.irecv, t, peer) for t, peer in recvs]
        if not ops_:
            return []
        if self._host_ordered:      # see __init__: the backend does not follow the stream
            self.ops.sync()
        return dist.batch_isend_irecv(ops_)

    def _all_ints(self, values):
        """Every rank's list of small non-negative ints, as [rank][i] (one all-reduce of a
        zero-padded vector: works with any communicator that can sum)."""
        k = len(values)
        if self._ints is None or self._ints.shape[0] != self.world * k:
            self._ints = self.ops.alloc((self.world * k,))
        t = self._ints
        self.ops.zero(t)
        t[self.rank * k:(self.rank + 1) * k] = self._as_tensor(values, t)
        if self._host_ordered:
            self.ops.sync()
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        flat = [int(round(v)) for v in t.tolist()]
        return [flat[r * k:(r + 1) * k] for r in range(self.world)]

    @staticmethod
    def _as_tensor(values, like):
        import torch
        return torch.tensor([float(v) for v in values], dtype=like.dtype, device=like.device)

    def flux_exchange_start(self, planes=FLUX_PLANES, counts=None):
        """Ship the flux deposited into my ghost rows to their owners.  counts = (send_up,
        send_down, recv_up, recv_down) rows, nearest the boundary first; default: all of them."""
        if counts is None:
            counts = (self.gu, self.gd, self._peer_ghost(self.up), self._peer_ghost(self.down))
        send_up, send_down, recv_up, recv_down = counts
        sends, recvs = [], []
        for name in planes:
            t = self.P[name]
            su, sd = self.stage[name]
            if self.up is not None:
                if send_up:
                    sends.append((t[self.r0 - send_up:self.r0], self.up))
                if recv_up:
                    recvs.append((su[0:recv_up], self.up))
            if self.down is not None:
                if send_down:
                    sends.append((t[self.r1:self.r1 + send_down], self.down))
                if recv_down:
                    recvs.append((sd[0:recv_down], self.down))
        self.halo_rows["flux"] += (send_up + send_down) * len(planes)
        self.halo_rows["full"] += (self.gu + self.gd) * len(planes)
        return self._exchange(sends, recvs), counts

    def flux_exchange_finish(self, started, planes=FLUX_PLANES):
        reqs, (send_up, send_down, recv_up, recv_down) = started
        for r in reqs:
            r.wait()
        for name in planes:
            t = self.P[name]
            su, sd = self.stage[name]
            if su is not None and recv_up:   # the up neighbour's lower ghost rows = my first owned rows
                self.ops.add(t[self.r0:self.r0 + recv_up], su[0:recv_up])
            if sd is not None and recv_down:
                self.ops.add(t[self.r1 - recv_down:self.r1], sd[0:recv_down])
            if self.up is not None and send_up:
                self.ops.zero(t[self.r0 - send_up:self.r0])
            if self.down is not None and send_down:
                self.ops.zero(t[self.r1:self.r1 + send_down])

    def field_exchange(self, layers_key="layers", counts=None):
        """Refresh the ghost rows of the fields the next step's particles read.  counts =
        (need_up, need_down, give_up, give_down): rows I want from / owe to each neighbour,
        nearest the boundary first; default: the whole ghost zones."""
        if counts is None:
            counts = (self.gu, self.gd, self._peer_ghost(self.up), self._peer_ghost(self.down))
        need_up, need_down, give_up, give_down = counts
        sends, recvs = [], []
        for name in FIELD_PLANES:
            t = self.P[layers_key if name == "layers" else name]
            if self.up is not None:
                if give_up:
                    sends.append((t[self.r0:self.r0 + give_up], self.up))
                if need_up:
                    recvs.append((t[self.r0 - need_up:self.r0], self.up))
            if self.down is not None:
                if give_down:
                    sends.append((t[self.r1 - give_down:self.r1], self.down))
                if need_down:
                    recvs.append((t[self.r1:self.r1 + need_down], self.down))
        self.halo_rows["field"] += (give_up + give_down) * len(FIELD_PLANES)
        self.halo_rows["full"] += (self._peer_ghost(self.up) + self._peer_ghost(self.down)) * len(FIELD_PLANES)
        for r in self._exchange(sends, recvs):
            r.wait()
        self.fresh_up, self.fresh_down = need_up, need_down

    # -- measured reach ----------------------------------------------------------
    def _reach(self, planes):
        """[rank] -> (rows above, rows below) its owned rows this launch's deposits got to."""
        up, down = self.ops.ghost_extent([self.P[n] for n in planes], self.r0, self.r1)
        return self._all_ints([up, down])

    def _too_deep(self, reach):
        """Did a launch, on any rank, get within a row of ghost rows that were not refreshed?
        (The cell record of ghost row d is made of rows d - 1 .. d + 1.)  The answer is the same
        on every rank: everybody knows everybody's reach and refreshed depth."""
        for r, (up, down) in enumerate(reach):
            f_up, f_down = self._fresh_all[r]
            x0, rows, r0, r1 = slab_layout(r, self.world, self.S, self.G)
            if (up >= f_up and f_up < r0) or (down >= f_down and f_down < rows - r1):
                return True
        return False

    def _refresh_all(self):
        """The prediction was too small: fetch the whole ghost zones of the fields as they stand
        (the cell phase of this step has not touched them yet)."""
        self.fallbacks += 1
        self.field_exchange("layers")
        self._fresh_all = [(slab_layout(r, self.world, self.S, self.G)[2],
                            slab_layout(r, self.world, self.S, self.G)[1] -
                            slab_layout(r, self.world, self.S, self.G)[3]) for r in range(self.world)]

    def _predict_need(self):
        """Ghost rows to refresh for the next step: as deep as the walks of the last steps got
        anywhere, a tenth more and ten rows on top (the reach moves by a row or two from step to step);
        everything while there is no history."""
        if not self.reach_hist:
            return self.gu, self.gd
        want = int(1.1 * max(self.reach_hist)) + 10
        if os.environ.get("SOIL_HALO_NEED"):    # tests: a prediction that is too small on purpose
            want = int(os.environ["SOIL_HALO_NEED"])
        return min(self.gu, want), min(self.gd, want)

    # -- one step ---------------------------------------------------------------
    def step(self, ev=None):
        ops, P = self.ops, self.P
        trim = self.trim
        ops.seed(self.rng, self.seed, self.step_index * self.N)
        ops.zero(self.remote0)
        early = ()          # flux planes whose halo is exchanged before the debris launch ends
        counts_f = counts_d = None
        if ev: ev.record(0)
        if hasattr(ops, "particles_pair") and not self.serial_particles and not trim:
            # the debris launch draws from a tensor of its own, seeded where the fluvial
            # launch leaves the shared one in the sequential order
            ops.seed(self.rng_debris, self.seed, self.step_index * self.N + 2)
            ops.particles_pair(P, self.rng, self.rng_debris, self.N, self.dom, self.scale,
                               self.param, self.remote0)
            if ev: ev.record(1)
        else:
            ops.particles_fluvial(P, self.rng, self.N, self.dom, self.scale, self.param,
                                  self.remote0)
            if trim:
                reach_f = self._reach(FLUX_FLUVIAL)
                if self._too_deep(reach_f):      # rare: repeat the launch on complete fields
                    self._refresh_all()
                    for name in FLUX_FLUVIAL:
                        ops.zero(P[name])
                    ops.zero(self.remote0)
                    ops.seed(self.rng, self.seed, self.step_index * self.N)
                    ops.particles_fluvial(P, self.rng, self.N, self.dom, self.scale, self.param,
                                          self.remote0)
                    reach_f = self._reach(FLUX_FLUVIAL)
                me = reach_f[self.rank]
                counts_f = (me[0], me[1],
                            reach_f[self.up][1] if self.up is not None else 0,
                            reach_f[self.down][0] if self.down is not None else 0)
            if ev: ev.record(1)
            if self.world > 1:
                # the fluvial flux is final: its halo travels, and is added, while the
                # debris launch runs
                with ops.fork_comm():
                    self.flux_exchange_finish(self.flux_exchange_start(FLUX_FLUVIAL, counts_f),
                                              FLUX_FLUVIAL)
                early = FLUX_FLUVIAL
            ops.particles_debris(P, self.rng, self.N, self.dom, self.scale, self.param,
                                 self.remote0)
            if trim:
                reach_d = self._reach(FLUX_DEBRIS)
                if self._too_deep(reach_d):
                    self._refresh_all()
                    for name in FLUX_DEBRIS:
                        ops.zero(P[name])
                    # the NaN walkers' debris deposits are entries 4..6 of remote0; the launch
                    # draws where the fluvial one left the streams (two draws per particle on)
                    ops.zero(self.remote0[4:8])
                    ops.seed(self.rng, self.seed, self.step_index * self.N + 2)
                    ops.particles_debris(P, self.rng, self.N, self.dom, self.scale, self.param,
                                         self.remote0)
                    reach_d = self._reach(FLUX_DEBRIS)
                me = reach_d[self.rank]
                counts_d = (me[0], me[1],
                            reach_d[self.up][1] if self.up is not None else 0,
                            reach_d[self.down][0] if self.down is not None else 0)
                self.reach_hist = (self.reach_hist + [max(max(a, b) for a, b in reach_f + reach_d)])[-4:]
        if ev: ev.record(2)
        if self.world == 1:
            ops.cells(P, self.dom, self.r0, self.r1, self.scale, self.param)
        else:
            # NaN walkers of the other ranks -> global cell (0,0) (8 floats, latency only)
            if self._host_ordered:
                ops.sync()
            self.dist.all_reduce(self.remote0)
            if self.rank == 0:
                ops.add_cell0(P, self.remote0)
            # rows whose flux is complete without the neighbours' contribution
            i0 = min(self.r1, self.r0 + (self._peer_ghost(self.up) if self.up is not None else 0))
            i1 = max(i0, self.r1 - (self._peer_ghost(self.down) if self.down is not None else 0))
            # 1. the rest of the flux halo (exposed: the bands below need it)
            late = tuple(n for n in FLUX_PLANES if n not in early)
            self.flux_exchange_finish(self.flux_exchange_start(late, counts_d if trim else None), late)
            ops.join_comm()                      # ... and the part that travelled early
            if ev: ev.record(4)                  # 2 -> 4: flux halo not hidden by the debris launch
            # 2. the bands next to the neighbours first: they are what the neighbours' ghost
            #    rows get
            ops.cells(P, self.dom, self.r0, i0, self.scale, self.param)
            ops.cells(P, self.dom, i1, self.r1, self.scale, self.param)
            # 3. the field halo travels while the interior rows are computed (they are
            #    G rows away from anything the exchange reads or writes)
            counts = None
            if trim:     # as deep as next step's walks are expected to get; everybody says what it wants
                need = self._predict_need()
                wants = self._all_ints(list(need))
                counts = (need[0], need[1],
                          wants[self.up][1] if self.up is not None else 0,
                          wants[self.down][0] if self.down is not None else 0)
                self._fresh_all = [tuple(w) for w in wants]
            with ops.fork_comm():
                self.field_exchange("layers_next", counts)
            ops.cells(P, self.dom, i0, i1, self.scale, self.param)
            if ev: ev.record(5)                  # 5 -> 3: field halo not hidden by the interior rows
            ops.join_comm()
        if ev: ev.record(3)
        P["layers"], P["layers_next"] = P["layers_next"], P["layers"]
        self.step_index += 1

    # -- bench plumbing ----------------------------------------------------------
    def sync(self):
        self.ops.sync()

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def shutdown(self):
        """Tear the process group down (only when it is torch.distributed itself)."""
        d = self.dist
        if hasattr(d, "is_initialized") and d.is_initialized():
            self.sync()
            d.barrier()
            d.destroy_process_group()

    def max_over_ranks(self, value):
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64,
                         device=getattr(self.ops, "device", "cpu"))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())


# ---- flow accumulation: replicas, realisations sharded ------------------------------

def multiflow(height, source, K, T, edge=1, seed=0, comm=None, rank=None, world=None,
              local_sum=None):
    """Stochastic multiple-flow accumulation (example/dem_multiflow.py:43-49) over all
    ranks.  `accumulate` does not shard — its pointer jumps span the whole grid
     — so every rank holds the full DEM and computes the realisations
    k = rank, rank + world, ... < K; one all-reduce(sum) of the float64 mean plane
    (128 MiB at 4096^2) combines them.

    `height`, `source`: silt.gpu float32 tensors (H, W), the same on every rank.
    Returns a torch float64 tensor (H, W) holding the mean on every rank.
    `local_sum(first, stride) -> torch tensor` replaces the HIP back-end in the
    CPU (gloo) tests."""
    if comm is None:
        import torch.distributed as comm
    if world is None:
        live = comm.is_available() and comm.is_initialized()
        rank, world = (comm.get_rank(), comm.get_world_size()) if live else (0, 1)
    if local_sum is None:
        from . import soil

        def local_sum(first, stride):
            out = soil.multiflow(height, source, K, T, edge, seed, first=first, stride=stride)
            return out.view_torch()
    total = local_sum(rank, world)
    if world > 1:
        comm.all_reduce(total, op=comm.ReduceOp.SUM)
    return total
