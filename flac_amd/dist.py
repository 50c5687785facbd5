"""flac_amd.dist -- multi-GPU sharding of a frame-parallel encode job.

Frames are independent, so a corpus shards by contiguous frame ranges with NO data-path collective:
rank r encodes frames [r*F/W, (r+1)*F/W) with the frame numbers fixed up front.  The only exchange is
the ORDERED gather of the variable-length bitstream to one rank:
    all_gather(byte totals, frame counts) -> exclusive scan -> point-to-point payload sends
i.e. a Gatherv built from RCCL send/recv over xGMI (works identically over gloo on CPU, which is how
tests/test_dist_cpu.py covers it).

Three forms of it:
  ordered_gather   one shot (the end of a corpus job): one tiny all_gather, ONE host read of the sizes, one batch of
                   point-to-point transfers into a buffer the destination may preallocate.
  GatherPipeline   steady state (bench.py, a corpus encoded in super-batches): the steps of a WINDOW are encoded
                   back to back with no host involvement, their sizes are exchanged and read once per window, the
                   window's transfers are posted in one batch on a communication stream, and the next window's
                   encodes run meanwhile.  The destination's own frames are encoded straight into the gathered
                   stream (its payload comes first), so nothing of its own is ever copied.
  HostShmGather    the variant that scales past one GPU's links: every rank copies its shard over ITS OWN PCIe link
                   into one shared pinned host buffer at the scanned offset; only the byte totals cross between
                   ranks.  No rank-0 funnel (7 peers x 50 GB/s into one GPU at -8 rates).
"""
import mmap
import os
import secrets
import socket
import sys

import torch
import torch.distributed as dist


# ------------------------------------------------------------------------------------------------------------------
# one call, N workers (the reference: FLAC__stream_encoder_set_num_threads, stream_encoder.c:2151 -- one call starts the
# workers, frames come out in order, :3530-3574).  A job's command line says `--gpus N`; if it was not started under a
# launcher already it starts its N ranks itself.
# ------------------------------------------------------------------------------------------------------------------
EXIT_BAD_WORLD = 3           # --gpus N disagrees with the launcher's world size, or fewer than N devices are visible


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_command(nproc, script, argv, port=None, python=None, module=False):
    """The command that runs `script argv...` (module=True: `-m script`) as `nproc` ranks of one node, one rank per GPU
    (torch.distributed.run: RANK, LOCAL_RANK, WORLD_SIZE, MASTER_* in every rank's environment) -- exactly what the round
    driver types for N > 1."""
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(nproc)),
            "--master-addr", "127.0.0.1", "--master-port", str(int(port or free_port()))] + (["-m"] if module else []) + [script] + list(argv)


def visible_devices():
    try:
        return int(torch.cuda.device_count()) if torch.cuda.is_available() else 0
    except Exception:
        return 0


def ensure_ranks(nproc, script, argv, need_devices=True, env=None, _exec=None, module=False):
    """Call first thing in a job's main(): returns (rank, local_rank, world) of THIS process, having made sure the job runs
    as `nproc` ranks.
      * started under a launcher (WORLD_SIZE set): the launcher's world size must equal `nproc` when `nproc` is given --
        otherwise the process ends with EXIT_BAD_WORLD (a line that says n_gpus: 8 must come from 8 ranks);
      * not under a launcher and nproc in (None, 1): one rank, no process group;
      * not under a launcher and nproc > 1: refuse (EXIT_BAD_WORLD) when fewer than nproc devices are visible
        (need_devices), else REPLACE this process by the launcher (launch_command) -- stdout, stderr, signals and the exit
        code are the launcher's, so rank 0's one JSON line is the command's one line.
    `_exec` is for the tests (a function taking the command instead of os.execvpe)."""
    env = os.environ if env is None else env
    if "WORLD_SIZE" in env:
        world = int(env["WORLD_SIZE"])
        if nproc is not None and int(nproc) != world:
            sys.stderr.write("flac_amd.dist: --gpus %d but the launcher started %d ranks (WORLD_SIZE)\n" % (int(nproc), world))
            raise SystemExit(EXIT_BAD_WORLD)
        return int(env.get("RANK", "0")), int(env.get("LOCAL_RANK", "0")), world
    # (FLAC_AMD_FORCE_LAUNCH=1: go through the launcher even for one rank -- the tests' way to run the self-launch path, exec and
    #  all, on a box with a single GPU)
    if (nproc is None or int(nproc) <= 1) and not (env.get("FLAC_AMD_FORCE_LAUNCH") == "1" and nproc is not None):
        return 0, 0, 1
    nproc = int(nproc)
    if need_devices:
        have = visible_devices()
        if have < nproc:
            sys.stderr.write("flac_amd.dist: --gpus %d but %d HIP device(s) visible\n" % (nproc, have))
            raise SystemExit(EXIT_BAD_WORLD)
    cmd = launch_command(nproc, script, argv, module=module)
    child_env = dict(env)
    child_env.pop("FLAC_AMD_FORCE_LAUNCH", None)
    child_env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # the host driver only supports dmabuf IPC (RCCL across processes)
    child_env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // nproc)))
    sys.stdout.flush()
    sys.stderr.flush()
    if _exec is not None:
        return _exec(cmd, child_env)
    os.execvpe(cmd[0], cmd, child_env)


def check_world(nproc, group=None):
    """After init_process_group: the group really has the size the command line asked for."""
    world = dist.get_world_size(group)
    if nproc is not None and world != int(nproc):
        sys.stderr.write("flac_amd.dist: the process group has %d ranks, the job asked for %d\n" % (world, int(nproc)))
        raise SystemExit(EXIT_BAD_WORLD)
    return world


class Watchdog:
    """Context manager around work whose failure must not take a finished result with it (bench.py: the side figures behind the
    multi-rank line).  A thread calls on_fail(reason) -- which is expected not to return: print what there is, os._exit -- when
    the block has not finished after `timeout` seconds, or when SIGTERM arrives (the launcher's way of ending the other ranks
    after one has died).  The signal is seen through the interpreter's wake-up descriptor, written by the C-level handler the
    moment the signal arrives: a main thread that sits in a collective or a device synchronisation never gets to run a Python
    handler.  Main thread only (signal.set_wakeup_fd)."""

    def __init__(self, timeout, on_fail, what="the guarded block"):
        self.timeout, self.on_fail, self.what = float(timeout), on_fail, what

    def __enter__(self):
        import signal
        import threading
        self._signal = signal
        self._rp, self._wp = os.pipe()
        os.set_blocking(self._wp, False)
        self._old_fd = signal.set_wakeup_fd(self._wp, warn_on_full_buffer=False)
        self._old_term = signal.signal(signal.SIGTERM, lambda signum, frame: None)      # (a Python-level handler puts the C-level one in place)
        self._done = threading.Event()
        self._thread = threading.Thread(target=self._watch, daemon=True)
        self._thread.start()
        return self

    def _watch(self):
        import select
        import time
        deadline = time.monotonic() + self.timeout
        while not self._done.is_set():
            left = deadline - time.monotonic()
            if left <= 0:
                self.on_fail("%s did not finish within %d s" % (self.what, int(self.timeout)))
                return
            ready, _, _ = select.select([self._rp], [], [], min(left, 0.5))
            if ready and not self._done.is_set():
                got = os.read(self._rp, 64)
                if bytes([self._signal.SIGTERM]) in got:
                    self.on_fail("signal %d (another rank failed?) during %s" % (int(self._signal.SIGTERM), self.what))
                    return

    def __exit__(self, et, ev, tb):
        self._done.set()
        self._signal.set_wakeup_fd(self._old_fd)
        self._signal.signal(self._signal.SIGTERM, self._old_term)
        self._thread.join(timeout=2.0)
        os.close(self._rp)
        os.close(self._wp)
        return False


def shard_range(nframes, world, rank):
    """Contiguous frame range [lo, hi) of `rank`; earlier ranks take the remainder."""
    base, rem = divmod(nframes, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _exchange(meta, group):
    """all_gather of a small int64 vector; returns the [world, len(meta)] table on the HOST (the one host sync)."""
    world = dist.get_world_size(group)
    flat = torch.empty(world * meta.numel(), dtype=torch.int64, device=meta.device)
    dist.all_gather_into_tensor(flat, meta.contiguous(), group=group)
    return flat.cpu().view(world, meta.numel())


def ordered_gather(payload, nbytes, frame_bytes, group=None, dst=0, out=None, out_frame_bytes=None):
    """Gather variable-length encoded shards to `dst` in rank order.

    payload     uint8 tensor (device or CPU) whose first `nbytes` bytes are this rank's frames
    nbytes      int, or a one-element integer tensor on payload's device (e.g. the engine's byte total: no host read here)
    frame_bytes uint32/int32 tensor [nframes_local] with the length of each local frame
    out / out_frame_bytes  (dst only, optional) preallocated uint8 / int64 buffers at least as large as the result
    Returns on dst: (stream uint8 tensor [total], all_frame_bytes int64 tensor [total frames]) -- views of `out`
    when given; on other ranks: (None, None).
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = payload.device
    nfr = int(frame_bytes.numel())
    if torch.is_tensor(nbytes):
        meta = torch.stack([nbytes.reshape(-1)[0].to(torch.int64), torch.tensor(nfr, dtype=torch.int64, device=dev)])
    else:
        meta = torch.tensor([int(nbytes), nfr], dtype=torch.int64, device=dev)
    table = _exchange(meta, group)
    sizes = [int(v) for v in table[:, 0]]
    counts = [int(v) for v in table[:, 1]]
    fb = frame_bytes.to(torch.int64)
    if rank == dst:
        total, totalf = sum(sizes), sum(counts)
        stream = out[:total] if out is not None else torch.empty(total, dtype=torch.uint8, device=dev)
        allfb = out_frame_bytes[:totalf] if out_frame_bytes is not None else torch.empty(totalf, dtype=torch.int64, device=dev)
        ops, off, foff = [], 0, 0
        for r in range(world):
            if r == rank:
                if stream.data_ptr() + off != payload.data_ptr():      # (already in place when the caller encoded into `out`)
                    stream[off:off + sizes[r]].copy_(payload[:sizes[r]])
                allfb[foff:foff + counts[r]].copy_(fb)
            else:
                if sizes[r]:
                    ops.append(dist.P2POp(dist.irecv, stream[off:off + sizes[r]], r, group))
                if counts[r]:
                    ops.append(dist.P2POp(dist.irecv, allfb[foff:foff + counts[r]], r, group))
            off += sizes[r]
            foff += counts[r]
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return stream, allfb
    ops = []
    if sizes[rank]:
        ops.append(dist.P2POp(dist.isend, payload[:sizes[rank]].contiguous(), dst, group))
    if nfr:
        ops.append(dist.P2POp(dist.isend, fb.contiguous(), dst, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return None, None


class GatherPipeline:
    """Steady-state ordered gather for a job that encodes step after step (see the module docstring).

    Per step every rank produces up to `cap_bytes` of frames and `nframes` frame lengths.  Usage, identically on all
    ranks (the destination must be rank 0 of the group: its payload is the head of every gathered stream):
        gp = GatherPipeline(cap_bytes, nframes, device, window=4)
        for k in range(steps):
            gp.wait_slot_free(k)                 # device-side wait on gp.enc_stream, never a host wait
            out, fb, total = gp.slot(k)          # where step k must be encoded to (device tensors)
            ... enqueue the encode of step k on gp.enc_stream ...
            gp.step_done(k)                      # end of a window: gathers the PREVIOUS window
        gp.flush()                               # the windows still pending
    The gather of window w is issued only after window w+1 has been enqueued, so the one host read it needs (the
    sizes) never leaves the encode stream without work.  On dst, gp.gathered(k) is step k's stream in rank order.
    CPU tensors (gloo) are supported for the tests: streams and events are then no-ops.
    """

    def __init__(self, cap_bytes, nframes, device, window=4, group=None, dst=0):
        self.group, self.dst = group, dst
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        if dst != 0:
            raise ValueError("GatherPipeline: the destination must be rank 0 of the group")
        self.cap, self.nframes, self.K = int(cap_bytes), int(nframes), max(1, int(window))
        self.dev = torch.device(device)
        self.cuda = self.dev.type == "cuda"
        self.is_dst = self.rank == dst
        self.backend = dist.get_backend(group)
        K, W = self.K, self.world
        nslots = 2 * K                                     # two windows in flight: one being encoded, one being gathered
        # dst: a step's gathered stream, its own payload first (encoded in place); others: just their payload
        self.span = self.cap * (W if self.is_dst else 1)
        self.buf = torch.empty(nslots * self.span, dtype=torch.uint8, device=self.dev)
        self.fb = torch.empty(nslots * self.nframes, dtype=torch.int32, device=self.dev)
        self.totals = torch.zeros(nslots, dtype=torch.int64, device=self.dev)
        self.allfb = torch.empty(nslots * W * self.nframes, dtype=torch.int32, device=self.dev) if self.is_dst else None
        self.sizes = [None] * nslots                       # dst: per-rank byte counts of the step in that slot (host ints)
        if self.cuda:
            self.enc_stream, self.comm_stream = torch.cuda.Stream(self.dev), torch.cuda.Stream(self.dev)
            self.win_done = [torch.cuda.Event() for _ in range(2)]     # the window's encodes are finished
            self.win_free = [torch.cuda.Event() for _ in range(2)]     # the window's buffers were read by the gather
        self.pending = []                                  # windows enqueued but not yet gathered: (first step, steps)
        self.enqueued = 0
        self.gathers = 0
        self.host_syncs = 0
        self.win_log = []                                  # per gathered window: [first step, steps, bytes moved by this rank, event0, event1]

    def window_stats(self):
        """Per gathered window, after the communication stream was synchronised: first step, steps, the bytes this rank
        received (dst) or sent (others), and the milliseconds its transfers took on the communication stream (None on CPU)."""
        out = []
        for k0, n, nbytes, e0, e1 in self.win_log:
            out.append({"first_step": k0, "steps": n, "bytes": nbytes, "ms": round(e0.elapsed_time(e1), 4) if e0 is not None else None})
        return out

    def _slot(self, k):
        return k % (2 * self.K)

    def slot(self, k):
        s = self._slot(k)
        return (self.buf[s * self.span: s * self.span + self.cap], self.fb[s * self.nframes:(s + 1) * self.nframes], self.totals[s:s + 1])

    def wait_slot_free(self, k):
        """Device-side: the encode stream waits until the gather that last read step k's window is through."""
        w = k // self.K
        if self.cuda and k % self.K == 0 and w >= 2:
            self.enc_stream.wait_event(self.win_free[w % 2])

    def _close_window(self, k0, n):
        if self.cuda:
            self.win_done[(k0 // self.K) % 2].record(self.enc_stream)
        self.pending.append((k0, n))

    def step_done(self, k):
        """Call after step k was enqueued."""
        self.enqueued = k + 1
        if (k + 1) % self.K == 0:
            self._close_window(k + 1 - self.K, self.K)
            if len(self.pending) > 1:
                self._gather_window(*self.pending.pop(0))

    def flush(self):
        part = self.enqueued % self.K
        if part:
            self._close_window(self.enqueued - part, part)
        while self.pending:
            self._gather_window(*self.pending.pop(0))

    def _gather_window(self, k0, n):
        import contextlib
        w = (k0 // self.K) % 2
        s0 = self._slot(k0)
        with (torch.cuda.stream(self.comm_stream) if self.cuda else contextlib.nullcontext()):
            if self.cuda:
                self.comm_stream.wait_event(self.win_done[w])
            # the ONE host read of this window (the next window's encodes are already queued on the encode stream)
            table = _exchange(self.totals[s0:s0 + n], self.group)          # [world, n]
            self.host_syncs += 1
            ops = []
            moved = 0
            for j in range(n):
                s = s0 + j
                sz = [int(v) for v in table[:, j]]
                moved += sum(sz[1:]) if self.is_dst else sz[self.rank]
                if self.is_dst:
                    self.sizes[s] = sz
                    base, off = s * self.span, sz[0]                         # rank 0's own payload is already there
                    for r in range(1, self.world):
                        if sz[r]:
                            ops.append(dist.P2POp(dist.irecv, self.buf[base + off: base + off + sz[r]], r, self.group))
                        a = (s * self.world + r) * self.nframes
                        ops.append(dist.P2POp(dist.irecv, self.allfb[a:a + self.nframes], r, self.group))
                        off += sz[r]
                else:
                    if sz[self.rank]:
                        ops.append(dist.P2POp(dist.isend, self.buf[s * self.span: s * self.span + sz[self.rank]], self.dst, self.group))
                    ops.append(dist.P2POp(dist.isend, self.fb[s * self.nframes:(s + 1) * self.nframes], self.dst, self.group))
            e0 = e1 = None
            if self.cuda:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(self.comm_stream)
            if ops:
                for wk in dist.batch_isend_irecv(ops):
                    wk.wait()                     # stream-ordered on CUDA (the host does not block); blocking on gloo
            if self.cuda:
                e1.record(self.comm_stream)
                self.win_free[w].record(self.comm_stream)
            self.win_log.append([k0, n, moved, e0, e1])
        self.gathers += 1

    def gathered(self, k):
        """dst only, once the window of step k was gathered and the communication stream synchronised by the caller:
        (stream bytes of step k in rank order, [per-rank byte counts], frame lengths [world, nframes])."""
        s = self._slot(k)
        sz = self.sizes[s]
        fbs = self.allfb[s * self.world * self.nframes:(s + 1) * self.world * self.nframes].view(self.world, self.nframes).clone()
        fbs[self.rank] = self.fb[s * self.nframes:(s + 1) * self.nframes]
        return self.buf[s * self.span: s * self.span + sum(sz)], sz, fbs


def _shm_create(capacity, group, prefix="flacgpu_gather"):
    """Rank 0 creates the shared file under /dev/shm -- unpredictable name, O_EXCL | O_NOFOLLOW, mode 0600, so that no other
    local user can plant a link there or read the stream -- and tells the others its name; every rank maps it."""
    rank = dist.get_rank(group)
    names = [None]
    if rank == 0:
        path = "/dev/shm/%s_%d_%s" % (prefix, os.getpid(), secrets.token_hex(8))
        fd = os.open(path, os.O_RDWR | os.O_CREAT | os.O_EXCL | os.O_NOFOLLOW, 0o600)
        os.ftruncate(fd, capacity)
        names[0] = path
    dist.broadcast_object_list(names, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    path = names[0]
    if rank != 0:
        fd = os.open(path, os.O_RDWR | os.O_NOFOLLOW)
    mm = mmap.mmap(fd, capacity)
    return path, fd, mm


def _host_register(host):
    try:
        return int(torch.cuda.cudart().cudaHostRegister(host.data_ptr(), host.numel(), 0)) == 0
    except Exception:
        return False


class HostShmGather:
    """Ordered gather through ONE pinned host buffer shared by the ranks of a node (POSIX shared memory, registered
    with the HIP runtime by every rank): rank r copies its payload device -> host at the exclusive scan of the byte
    totals.  The transfers run on every GPU's own PCIe link at once.  CPU tensors (the gloo tests) copy plainly."""

    def __init__(self, capacity_bytes, group=None):
        self.group = group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.capacity = int(capacity_bytes)
        self.path, self.fd, self.mm = _shm_create(self.capacity, group)
        self.host = torch.frombuffer(self.mm, dtype=torch.uint8)
        self.registered = torch.cuda.is_available() and _host_register(self.host)
        dist.barrier(group)

    def gather(self, payload, nbytes, stream=None):
        """Every rank: copy payload[:nbytes] to the shared buffer at its scanned offset.  Returns (offset, sizes list);
        the data is complete on all ranks after the barrier this call ends with."""
        dev = payload.device
        meta = nbytes.reshape(-1)[:1].to(torch.int64) if torch.is_tensor(nbytes) else torch.tensor([int(nbytes)], dtype=torch.int64, device=dev)
        table = _exchange(meta, self.group)
        sizes = [int(v) for v in table[:, 0]]
        off = sum(sizes[:self.rank])
        if off + sizes[self.rank] > self.capacity:
            raise RuntimeError("HostShmGather: %d bytes do not fit the shared buffer" % sum(sizes))
        if dev.type == "cuda":
            with torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream(dev)):
                self.host[off:off + sizes[self.rank]].copy_(payload[:sizes[self.rank]], non_blocking=self.registered)
            torch.cuda.current_stream(dev).synchronize() if stream is None else stream.synchronize()
        else:
            self.host[off:off + sizes[self.rank]].copy_(payload[:sizes[self.rank]])
        dist.barrier(self.group)
        return off, sizes

    def close(self):
        if self.registered:
            torch.cuda.cudart().cudaHostUnregister(self.host.data_ptr())
            self.registered = False
        self.host = None
        try:
            self.mm.close()
        except BufferError:
            pass
        os.close(self.fd)
        dist.barrier(self.group)
        if self.rank == 0 and os.path.exists(self.path):
            os.unlink(self.path)


class HostShmPipeline(GatherPipeline):
    """GatherPipeline whose destination is the shared pinned host buffer of HostShmGather instead of rank 0's HBM: same
    calls, same windows, same single host read of the sizes per window -- but a window's transfers are device -> host copies,
    every rank over its own PCIe link, to (slot base + exclusive scan of the step's byte totals).  For a corpus whose stream
    must end in host memory anyway, and the diagnostic twin of the RCCL funnel (bench.py --gather hostshm).
    host_cap_bytes: room per rank and step in the host buffer (the slot of a step is world * host_cap_bytes)."""

    def __init__(self, cap_bytes, nframes, device, window=4, group=None, host_cap_bytes=None):
        self.hcap = int(host_cap_bytes or cap_bytes)
        super().__init__(cap_bytes, nframes, device, window=window, group=group, dst=0)
        nslots = 2 * self.K
        # (every rank keeps only its own payload on the device: the base class sized rank 0's slots for the whole stream)
        if self.is_dst and self.world > 1:
            self.span = self.cap
            self.buf = torch.empty(nslots * self.span, dtype=torch.uint8, device=self.dev)
        self.hspan = self.world * self.hcap
        self.path, self.fd, self.mm = _shm_create(nslots * self.hspan, group, "flacgpu_pipe")
        self.host = torch.frombuffer(self.mm, dtype=torch.uint8)
        self.registered = self.cuda and _host_register(self.host)
        self.hfb = None
        self.all_sizes = [None] * nslots                    # every rank knows every rank's byte counts of the step in a slot
        self.backend = "hostshm+" + self.backend
        dist.barrier(group)

    def _gather_window(self, k0, n):
        import contextlib
        w = (k0 // self.K) % 2
        s0 = self._slot(k0)
        with (torch.cuda.stream(self.comm_stream) if self.cuda else contextlib.nullcontext()):
            if self.cuda:
                self.comm_stream.wait_event(self.win_done[w])
            table = _exchange(self.totals[s0:s0 + n], self.group)          # [world, n]: the ONE host read of this window
            self.host_syncs += 1
            e0 = e1 = None
            if self.cuda:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(self.comm_stream)
            moved = 0
            for j in range(n):
                s = s0 + j
                sz = [int(v) for v in table[:, j]]
                self.all_sizes[s] = sz
                if sum(sz) > self.hspan:
                    raise RuntimeError("HostShmPipeline: a step's %d bytes do not fit its slot of %d" % (sum(sz), self.hspan))
                off = s * self.hspan + sum(sz[:self.rank])
                mine = sz[self.rank]
                if mine:
                    self.host[off:off + mine].copy_(self.buf[s * self.span: s * self.span + mine], non_blocking=self.registered)
                moved += mine
            if self.cuda:
                e1.record(self.comm_stream)
                self.win_free[w].record(self.comm_stream)
            self.win_log.append([k0, n, moved, e0, e1])
        self.gathers += 1

    def gathered(self, k):
        """Any rank, once every rank has synchronised its communication stream and a barrier was passed: (step k's stream in
        rank order -- a view of the shared host buffer --, [per-rank byte counts], this rank's frame lengths)."""
        s = self._slot(k)
        sz = self.all_sizes[s]
        return self.host[s * self.hspan: s * self.hspan + sum(sz)], sz, self.fb[s * self.nframes:(s + 1) * self.nframes]

    def close(self):
        if self.registered:
            torch.cuda.cudart().cudaHostUnregister(self.host.data_ptr())
            self.registered = False
        self.host = None
        try:
            self.mm.close()
        except BufferError:
            pass
        os.close(self.fd)
        dist.barrier(self.group)
        if self.rank == 0 and os.path.exists(self.path):
            os.unlink(self.path)
