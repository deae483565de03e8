"""Multi-GPU plumbing for the prediction path: independent trajectories shard by contiguous batch rows,
weights are replicated, and the ONLY collective is one all-gather of the per-sample metric rows per batch
(mirrors ``accelerator.gather`` of mse/psnr/ssim/lpips at /root/reference/train_gpt.py:476-479).
One process per GPU; ``torch.distributed`` backend "nccl" is RCCL on ROCm (xGMI), "gloo" in the CPU tests."""
import os
import threading

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """-> (rank, world_size, local_rank).  Initialises the default process group when WORLD_SIZE > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or os.environ.get("IVG_FORCE_COLLECTIVE") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_rows(n_rows, rank, world):
    """Contiguous row range [lo, hi) of rank `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_rows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_metric_rows(local_rows, total_rows=None):
    """All-gather per-sample metric rows [B_local, n_metrics] -> [B_total, n_metrics] in rank order on every rank.
    Ranks may hold different row counts (uneven shard): rows are padded to the max and trimmed after the gather."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local_rows
    world = dist.get_world_size()
    n = torch.tensor([local_rows.shape[0]], device=local_rows.device, dtype=torch.int64)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    mx = max(counts)
    pad = local_rows
    if local_rows.shape[0] < mx:
        pad = torch.cat([local_rows, local_rows.new_zeros(mx - local_rows.shape[0], *local_rows.shape[1:])], 0)
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad.contiguous())
    rows = torch.cat([o[:c] for o, c in zip(out, counts)], 0)
    if total_rows is not None:
        assert rows.shape[0] == total_rows
    return rows


def gather_metric_rows_even(local_rows, force_collective=False):
    """Fast path when every rank holds the same number of rows: one ncclAllGather of [B_local, n_metrics].
    ``force_collective`` (or IVG_FORCE_COLLECTIVE=1): issue the collective even in a 1-rank group -- how a single-GPU lease
    exercises the RCCL path the N-GPU run takes."""
    force = force_collective or os.environ.get("IVG_FORCE_COLLECTIVE") == "1"
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return local_rows
    out = local_rows.new_empty(dist.get_world_size() * local_rows.shape[0], *local_rows.shape[1:])
    dist.all_gather_into_tensor(out, local_rows.contiguous())
    return out


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value, device, force_collective=False):
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force_collective):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


class LocalAccelerator:
    """The slice of HF ``accelerate.Accelerator`` the reference's eval loop uses (train_gpt.py:321-512: ``device``,
    ``num_processes``, ``is_main_process``, ``is_local_main_process``, ``gather``, ``unwrap_model``, ``log``,
    ``wait_for_everyone``) over ``torch.distributed`` -- backend "nccl" (RCCL over xGMI) on the GPUs, "gloo" in the CPU tests.
    ``force_collective=True`` issues the collectives even in a 1-rank group (how a single-GPU lease exercises the RCCL path)."""

    def __init__(self, device=None, force_collective=None):
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        self.force_collective = os.environ.get("IVG_FORCE_COLLECTIVE") == "1" if force_collective is None else force_collective
        self.logged = []

    @property
    def num_processes(self):
        return dist.get_world_size() if dist.is_initialized() else 1

    @property
    def process_index(self):
        return dist.get_rank() if dist.is_initialized() else 0

    @property
    def is_main_process(self):
        return self.process_index == 0

    @property
    def is_local_main_process(self):
        return int(os.environ.get("LOCAL_RANK", "0")) == 0

    def gather(self, tensor):
        """Concatenation over ranks along dim 0, on every rank (``accelerator.gather``; equal shapes on all ranks)."""
        if not dist.is_initialized() or (dist.get_world_size() == 1 and not self.force_collective):
            return tensor
        t = tensor.contiguous()
        if t.dim() == 0:
            t = t[None]
        out = t.new_empty(dist.get_world_size() * t.shape[0], *t.shape[1:])
        dist.all_gather_into_tensor(out, t)
        return out

    def unwrap_model(self, model):
        return getattr(model, "module", model)

    def wait_for_everyone(self):
        if dist.is_initialized() and (dist.get_world_size() > 1 or self.force_collective):
            dist.barrier()

    def log(self, values, step=None):
        self.logged.append((step, dict(values)))


class Turnstile:
    """Issues work in ticket order, whatever thread it comes from: with several batches in flight per GPU (one host thread and one
    HIP stream per lane) the metric all-gathers of the lanes must reach RCCL in the SAME order on every rank -- collectives of one
    process group are matched by issue order.  Ticket = global step index; a lane's gather waits until all earlier steps' gathers
    have been issued (they are tiny and asynchronous on the device, so the wait is only for the other lane to REACH its gather)."""

    def __init__(self):
        self._cv = threading.Condition()
        self._turn = 0
        self._aborted = False

    def run(self, ticket, fn):
        with self._cv:
            self._cv.wait_for(lambda: self._turn == ticket or self._aborted)
            if self._aborted:
                raise RuntimeError("turnstile aborted: another lane failed before its turn")
        try:
            return fn()
        finally:
            with self._cv:
                self._turn += 1
                self._cv.notify_all()

    def reset(self, turn=0):
        with self._cv:
            self._turn, self._aborted = turn, False
            self._cv.notify_all()

    def abort(self):
        """A lane died: release every waiter (their run() raises) instead of leaving them waiting for a ticket that never comes."""
        with self._cv:
            self._aborted = True
            self._cv.notify_all()


class OrderedGatherer:
    """The metric all-gathers of several batches in flight, issued by ONE thread on ONE stream of its own, in global step order.
    A lane hands over its rows (``submit(step, rows)``: an event on the lane's stream marks them ready) and runs on: no lane stream
    ever waits for a collective, and the collective's stream waits only for the event of the step whose turn it is.  With the
    gathers issued from the lanes themselves (``Turnstile``) every lane stream waits for RCCL's stream, which runs the gathers of
    ALL lanes in step order -- the lanes are forced into lock-step completion (measured: 3,911 instead of 5,725 frames/s with four
    lanes, 1-rank RCCL group; profiles/r04_lanes.txt).  ``finish(n)`` returns the gathered rows of steps 0 .. n-1.
    Create it BEFORE the lane streams are used, so that its stream gets a hardware queue of its own."""

    def __init__(self, device, gather=None):
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None   # (CPU / gloo: the tests)
        self._gather = gather or gather_metric_rows_even
        self._cv = threading.Condition()
        self._pending, self._done, self._next, self._error, self._stop = {}, {}, 0, None, False
        self._thread = None

    def start(self, first=0):
        with self._cv:
            self._pending, self._done, self._next, self._error, self._stop = {}, {}, first, None, False
        self._thread = threading.Thread(target=self._run, daemon=True)
        self._thread.start()

    def submit(self, step, rows):
        ev = None
        if self.stream is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
        with self._cv:
            self._pending[step] = (rows, ev)
            self._cv.notify_all()

    def _run(self):
        try:
            if self.stream is not None:
                torch.cuda.set_device(self.device)
            while True:
                with self._cv:
                    self._cv.wait_for(lambda: self._stop or self._next in self._pending)
                    if self._stop and self._next not in self._pending:
                        return
                    rows, ev = self._pending.pop(self._next)
                if self.stream is not None:
                    with torch.cuda.stream(self.stream):
                        self.stream.wait_event(ev)
                        out = self._gather(rows)
                        rows.record_stream(self.stream)
                else:
                    out = self._gather(rows)
                with self._cv:
                    self._done[self._next] = out
                    self._next += 1
                    self._cv.notify_all()
        except Exception as e:   # surfaced by finish()
            with self._cv:
                self._error = e
                self._cv.notify_all()

    def finish(self, upto):
        """Blocks (host) until the gathers of all steps < ``upto`` have been ISSUED; -> {step: gathered rows} (device work may still be
        in flight on ``self.stream``: synchronise it, or the device, before reading)."""
        with self._cv:
            self._cv.wait_for(lambda: self._error is not None or self._next >= upto)
            self._stop = True
            self._cv.notify_all()
            err, done = self._error, dict(self._done)
        if self._thread is not None:
            self._thread.join()
        if err is not None:
            raise err
        return done

    def abort(self):
        with self._cv:
            self._stop = True
            self._pending.clear()
            self._cv.notify_all()


class PhaseGate:
    """Orders one KIND of phase across the batches in flight on a GPU: at most one lane's convolution phase (context encode, frame
    decode: MFMA-bound grids of thousands of workgroups) is on the device at a time, while the other lanes' rollouts (14.7 k short
    HBM- / latency-bound launches) run beside it.  Device-side ordering only: a lane entering a gated phase makes ITS stream wait
    for the event that closed the previous gated phase (whatever lane ran it), queues its kernels and records the next event -- the
    host threads never wait for the device, only for each other while one of them queues a phase (a few hundred launches).
    ``with gate.phase(stream): ...`` around the calls of the phase; ``gate = None`` callers skip it."""

    def __init__(self):
        self._lock = threading.Lock()
        self._last = None

    def phase(self, stream):
        return _GatedPhase(self, stream)


class _GatedPhase:
    def __init__(self, gate, stream):
        self.gate, self.stream = gate, stream

    def __enter__(self):
        self.gate._lock.acquire()
        try:
            if self.gate._last is not None:
                self.stream.wait_event(self.gate._last)
        except BaseException:
            self.gate._lock.release()   # __exit__ does not run when __enter__ raises: every other lane would wait forever
            raise
        return self

    def __exit__(self, *exc):
        try:
            ev = torch.cuda.Event()
            ev.record(self.stream)
            self.gate._last = ev
        finally:
            self.gate._lock.release()
        return False


def cu_masked_stream(device, cu_bits):
    """A HIP stream whose kernels only run on the compute units whose bits are set in ``cu_bits`` (iterable of CU indices, 256 on an
    MI355X): ``hipExtStreamCreateWithCUMask`` wrapped as a ``torch.cuda.ExternalStream``.  Used to give the MFMA-bound convolution
    phases and the HBM- / latency-bound rollouts of several batches in flight DISJOINT parts of the chip (bench.py --cu-split)."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    words = [0] * 8
    for b in cu_bits:
        words[b >> 5] |= 1 << (b & 31)
    arr = (C.c_uint32 * 8)(*words)
    h = C.c_void_p()
    with torch.cuda.device(device):
        rc = hip.hipExtStreamCreateWithCUMask(C.byref(h), 8, arr)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed: {rc}")
    return torch.cuda.ExternalStream(h.value, device=device)
