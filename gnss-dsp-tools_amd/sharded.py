"""PRN x Doppler grid sharded over one-process-per-GPU ranks (torch.distributed; backend "nccl" is
RCCL over xGMI on ROCm, "gloo" in the CPU tests).

The reference's only parallelism is one task per PRN through multiprocessing.Pool
(acquire-gps-l1.py:105-108).  Here every (PRN, Doppler) row is independent, so the Doppler grid is
cut into contiguous slices, one per rank: forward FFTs are not duplicated and every rank keeps all
code spectra.  The path has exactly ONE exchange step: an all-gather of the per-rank peak records
(16 B per (epoch, PRN)), followed by a merge that scans the shards in global Doppler order with
strict '>' so ties resolve to the lowest Doppler bin exactly like the reference's scan
(acquire-gps-l1.py:36-39).  A plain max all-reduce would lose that tie rule.
"""
import contextlib

import numpy as np

from . import acquire


def _stamp(x):
    """(tensor, version counter) of a sample tensor at launch time (None for non-torch inputs of the CPU tests)."""
    return (x, x._version) if hasattr(x, "_version") else None


def _check_unmodified(stamp):
    """The tie-safe merge re-evaluates near-tied shard winners in complex128 FROM THE SAMPLES, at wait() time: a caller that refills
    the sample tensor in place between launch and wait() would have step i's peaks resolved against step i+1's samples.  torch counts
    in-place writes per tensor, so that mistake is caught here instead of producing silently wrong locations."""
    if stamp is not None and stamp[0]._version != stamp[1]:
        raise RuntimeError("sharded search: the sample tensor was modified in place between launch and wait(); the deferred tie-safe merge "
                           "re-reads it -- keep x alive and unmodified until wait() returns (use a second buffer for the next step)")


def doppler_bounds(nd, world):
    """Contiguous, balanced slices of the Doppler grid: rank r owns [b[r], b[r+1])."""
    return [(r * nd) // world for r in range(world + 1)]


class ShardedSearch:
    def __init__(self, engine=None, group=None, local_fn=None, always_gather=False):
        """engine: acquire.Engine on this rank's GPU.  local_fn(name, x, items, dopplers_slice, blocks) ->
        peaks tensor [nepoch, nitems, 2] float64 replaces the engine (CPU tests drive the exchange/merge
        logic with it; the product path always passes an engine)."""
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.engine = engine
        self.local_fn = local_fn
        self.always_gather = always_gather          # run the collective + merge even for world_size 1 (tests)
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._lanes, self._lane_order = [], "cost"

    def enable_job_lanes(self, n=2, options=None, order="cost", cus=None):
        """Searches of several signals (search_jobs): run the jobs side by side on `n` contexts with their own HIP streams (this one
        plus n - 1 more), as the reference runs its searches side by side (Pool(32), acquire-gps-l1.py:105-108).  The large kernels
        of one job own the device while they run (the N = 16384 transforms hold a CU's whole register file), so this is not about
        sharing CUs between them: what runs under the next job's kernels is the END of a job -- the ragged last round of a long
        kernel's workgroups, the small latency-bound launches that close a search (Doppler scan, tie-safe re-evaluation, record
        writes).  Jobs are queued in descending cost (cells x blocks), job k on lane k mod n, so the longest tails are covered and
        the cheapest job closes the step.  Results are the same records in the same order.  `cus` limits the extra lanes to that many
        compute units (experiments: keeping the memory-bound searches on a few CUs next to the others was measured and is slower,
        profiles/r06_config5_clocks_and_overlap.log)."""
        if self.engine is None:
            raise RuntimeError("job lanes: needs an engine (the CPU test stand-in has no streams)")
        self.close_job_lanes()
        self._lane_order = order
        for _ in range(max(0, int(n) - 1)):
            eng = acquire.Engine(self.engine.device)
            for k, v in (options or {}).items():
                eng.set_option(k, int(v))
            own = acquire.MaskedStream(self.engine.device, cus)   # every CU by default, but a hardware queue of its own (see MaskedStream)
            self._lanes.append((eng, own.torch_stream, own))
        return self

    def close_job_lanes(self):
        for eng, st, own in getattr(self, "_lanes", []):
            st.synchronize()
            eng.close()
            own.close()
        self._lanes = []

    @staticmethod
    def _job_cost(job):
        name = job["family"][0] if job.get("family") else job["name"]
        sig = acquire._signals.get(name) if isinstance(name, str) else name
        nitems = sum(len(i) for i in job["items"]) if job.get("family") else len(job["items"])
        return float(nitems) * len(job["dopplers"]) * sig.nfft * int(job["blocks"]) * int(job["x"].shape[0])

    def partition(self, nd, nitems):
        """cut the Doppler grid while every rank still gets >= 4 bins (forward FFTs are not duplicated);
        otherwise cut the item list (each rank then repeats the forward transforms for its items)."""
        if self.world == 1 or nd >= 4 * self.world or nitems < self.world:
            return "doppler"
        return "items"

    def _launch(self, name, x, items, dopplers, blocks):
        """Local part of one search.  Returns (local peaks, finish) where finish(gathered [world, ...]) -> merged peaks."""
        import torch
        dopplers = np.ascontiguousarray(dopplers, dtype=np.float64)
        items = list(items)
        run = self.local_fn
        if run is None:
            # kernels, the collective and the merge must share one stream: RCCL orders itself against torch's
            # current stream only
            self.engine.use_torch_stream(x.device)
            run = self.engine.search_batch_dev
        if self.partition(len(dopplers), len(items)) == "doppler":
            b = doppler_bounds(len(dopplers), self.world)
            local = run(name, x, items, dopplers[b[self.rank]:b[self.rank + 1]], blocks)

            def finish(gathered):
                if gathered.is_cuda:
                    # tie-safe merge: slice winners within eps of each other are re-evaluated in complex128 (every rank holds x)
                    return self.engine.merge_peaks_tiesafe_dev(name, x, items, dopplers, blocks, gathered, b[:-1])
                return merge_peaks_host(gathered.numpy(), b[:-1])
            return local, finish
        # item split: every rank searches the whole Doppler grid for a contiguous slice of the items; slices are padded to
        # a common length with "nothing found" records so that one fixed-size all-gather still does the exchange
        ib = doppler_bounds(len(items), self.world)
        width = max(ib[r + 1] - ib[r] for r in range(self.world))
        mine = items[ib[self.rank]:ib[self.rank + 1]]
        nepoch = x.shape[0]
        local = torch.zeros((nepoch, width, 2), dtype=torch.float64, device=x.device)
        local.view(torch.int64)[:, :, 1] = -1                        # idx = d_index = -1 (two int32 in one 8-byte word)
        if mine:
            local[:, :len(mine)] = run(name, x, mine, dopplers, blocks)

        def finish(gathered):
            return torch.cat([gathered[r, :, :ib[r + 1] - ib[r]] for r in range(self.world)], dim=1).contiguous()
        return local, finish

    def _exchange(self, local, async_op=False):
        import torch
        flat = local.contiguous().view(-1)
        gathered = torch.empty(self.world * flat.numel(), dtype=local.dtype, device=local.device)
        work = self.dist.all_gather_into_tensor(gathered, flat, group=self.group, async_op=async_op)      # the ONE collective
        return gathered.view((self.world,) + tuple(local.shape)), work

    def _solo(self):
        return self.world == 1 and not (self.always_gather and self.dist.is_initialized())

    def search_batch(self, name, x, items, dopplers, blocks):
        """x: [nepoch, nsamp] complex64 tensor, identical on every rank (CUDA for the engine path).
        Returns the merged peaks tensor [nepoch, nitems, 2] (gacq_peak records, global Doppler index)
        on every rank."""
        local, finish = self._launch(name, x, items, dopplers, blocks)
        if self._solo():
            return local
        gathered, _ = self._exchange(local)
        return finish(gathered)

    def search_batch_async(self, name, x, items, dopplers, blocks):
        """Like search_batch, but the all-gather is issued asynchronously (RCCL runs it on its own stream once the local
        kernels are done) and the merge is deferred to PendingSearch.wait().  Launching the next search before waiting on
        the previous one overlaps the exchange of step i with the compute of step i+1.
        x must stay alive AND unmodified until wait() returns: the deferred tie-safe merge re-evaluates near-tied shard winners from
        the samples (wait() raises if torch saw an in-place write to x in between)."""
        local, finish = self._launch(name, x, items, dopplers, blocks)
        if self._solo():
            return PendingSearch(local, None, None, None)
        gathered, work = self._exchange(local, async_op=True)
        return PendingSearch(local, gathered, work, finish, _stamp(x))

    def search_jobs(self, jobs):
        """Cold-start style multi-constellation search (BASELINE config 5): `jobs` is a list of dicts
        {name, x, items, dopplers, blocks}.  Every job's Doppler grid is sliced over the ranks like search_batch, all
        local peaks are packed into ONE buffer and exchanged with a single all-gather, then merged per job.
        Returns the list of merged peak tensors (one per job, [nepoch, nitems, 2])."""
        return self.search_jobs_async(jobs, async_op=False).wait()

    def search_jobs_async(self, jobs, async_op=True):
        """search_jobs with the single all-gather issued asynchronously and the per-job merges deferred to
        PendingJobs.wait(): queueing the next step before waiting puts the exchange under the next step's kernels.
        Every job's x must stay alive and unmodified until wait() returns (see search_batch_async)."""
        import torch
        locals_, bounds, shapes = [None] * len(jobs), [None] * len(jobs), [None] * len(jobs)
        lanes = getattr(self, "_lanes", []) if (self.local_fn is None and len(jobs) > 1) else []
        order = list(range(len(jobs)))
        if lanes:
            if self._lane_order == "cost":
                order.sort(key=lambda i: -self._job_cost(jobs[i]))
            elif not isinstance(self._lane_order, str):
                order = list(self._lane_order)                   # an explicit queueing order (experiments)
            cur = torch.cuda.current_stream(jobs[0]["x"].device)
            for _, st, _ in lanes:
                st.wait_stream(cur)                              # the samples were produced on the caller's stream
        for k, ji in enumerate(order):
            job = jobs[ji]
            dop = np.ascontiguousarray(job["dopplers"], dtype=np.float64)
            b = doppler_bounds(len(dop), self.world)
            lo, hi = b[self.rank], b[self.rank + 1]
            lane = k % (len(lanes) + 1) if lanes else 0          # lane 0: this context on the caller's stream
            engine = lanes[lane - 1][0] if lane else self.engine
            with (torch.cuda.stream(lanes[lane - 1][1]) if lane else contextlib.nullcontext()):
                if self.local_fn is not None:
                    loc = self.local_fn(job["name"], job["x"], job["items"], dop[lo:hi], job["blocks"])
                elif job.get("family"):
                    # several signals sharing everything but their code tables (E1B + E1C): `family` = signal names, `items` = one
                    # item list per signal; one stacked signal, forward transforms shared
                    engine.use_torch_stream(job["x"].device)
                    loc = engine.search_family_batch_dev(job["family"], job["x"], job["items"], dop[lo:hi], job["blocks"])
                else:
                    engine.use_torch_stream(job["x"].device)
                    loc = engine.search_batch_dev(job["name"], job["x"], job["items"], dop[lo:hi], job["blocks"])
                loc = loc.contiguous()
            if lane:
                loc.record_stream(cur)                           # allocated on the lane's stream, consumed on the caller's
            locals_[ji], bounds[ji], shapes[ji] = loc, b, tuple(loc.shape)
        for _, st, _ in lanes:
            cur.wait_stream(st)                                  # join: the exchange / the caller reads every job's records
        if self._solo():
            return PendingJobs(self, locals_, None, None, shapes, bounds, jobs)
        flat = torch.cat([t.view(-1) for t in locals_])
        gathered = torch.empty(self.world * flat.numel(), dtype=flat.dtype, device=flat.device)
        work = self.dist.all_gather_into_tensor(gathered, flat, group=self.group, async_op=async_op)      # the ONE collective
        return PendingJobs(self, locals_, gathered.view(self.world, flat.numel()), work if async_op else None, shapes, bounds, jobs)

    def results(self, name, items, merged, dopplers):
        """Merged peaks -> per-epoch lists of the reference's (metric, code, doppler) tuples."""
        pk = merged.cpu().numpy() if hasattr(merged, "cpu") else np.asarray(merged)
        pk = pk.view(acquire.PEAK_DTYPE).reshape(-1, len(items))
        return [acquire.finalize(name, items, pk[e], dopplers) for e in range(pk.shape[0])]


class PendingSearch:
    """A sharded search whose exchange may still be in flight (ShardedSearch.search_batch_async)."""

    def __init__(self, local, gathered, work, finish, stamp=None):
        self.local, self.gathered, self.work, self.finish, self.stamp = local, gathered, work, finish, stamp

    def wait(self):
        """Merged peaks [nepoch, nitems, 2].  For RCCL, Work.wait() orders the current stream after the collective
        (it does not block the host), so the merge kernel can be queued right away."""
        if self.gathered is None:
            return self.local
        if self.work is not None:
            self.work.wait()
        _check_unmodified(self.stamp)
        return self.finish(self.gathered)


class PendingJobs:
    """A sharded multi-job search whose single exchange may still be in flight (ShardedSearch.search_jobs_async)."""

    def __init__(self, owner, locals_, gathered, work, shapes, bounds, jobs=None):
        self.owner, self.locals_, self.gathered, self.work, self.shapes, self.bounds = owner, locals_, gathered, work, shapes, bounds
        self.jobs = jobs
        self.stamps = [_stamp(j["x"]) for j in jobs] if (jobs and gathered is not None) else []

    def shards(self):
        """The un-merged exchange buffer [world, sum of job record counts * 2] (None on a single rank): what every rank
        received from every other rank."""
        if self.work is not None:
            self.work.wait()
            self.work = None
        return self.gathered

    def wait(self):
        if self.gathered is None:
            return self.locals_
        g = self.shards()
        for st in self.stamps:
            _check_unmodified(st)
        out, off = [], 0
        for k, (shp, b) in enumerate(zip(self.shapes, self.bounds)):
            cnt = int(np.prod(shp))
            part = g[:, off:off + cnt].contiguous().view((self.owner.world,) + shp)
            off += cnt
            if not part.is_cuda:
                out.append(merge_peaks_host(part.numpy(), b[:-1]))
                continue
            job, eng = self.jobs[k], self.owner.engine
            if job.get("family"):
                base, fam, _ = eng._family(job["family"], job["items"])
                out.append(eng.merge_peaks_tiesafe_dev(fam.sig, job["x"], fam.prns, job["dopplers"], job["blocks"], part, b[:-1], _signal=fam))
            else:
                out.append(eng.merge_peaks_tiesafe_dev(job["name"], job["x"], job["items"], job["dopplers"], job["blocks"], part, b[:-1]))
        return out


def merge_peaks_host(gathered, shard_d0):
    """numpy form of gacq_merge_peaks_dev for CPU tensors: [nshard, nepoch, nitems, 2] -> [nepoch, nitems, 2]."""
    import torch
    g = np.ascontiguousarray(gathered).view(acquire.PEAK_DTYPE).reshape(gathered.shape[0], -1)
    out = np.zeros(g.shape[1], dtype=acquire.PEAK_DTYPE)
    out["idx"] = -1
    out["d_index"] = -1
    for s in range(g.shape[0]):
        win = (g[s]["d_index"] >= 0) & (g[s]["metric"] > out["metric"])       # strict '>' in shard order
        out["metric"][win] = g[s]["metric"][win]
        out["idx"][win] = g[s]["idx"][win]
        out["d_index"][win] = g[s]["d_index"][win] + shard_d0[s]
    return torch.from_numpy(out.view(np.float64).reshape(gathered.shape[1:]).copy())
