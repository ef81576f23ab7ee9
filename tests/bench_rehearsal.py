"""The ranks of `python bench.py --gpus N --rehearse-cpu` (TEST INFRASTRUCTURE; may use oracle/ through tests/cpu_engine.py).

bench.py's launcher, gloo rendezvous and timed-step protocol (barrier + max-over-ranks around exactly K steps) are the product's;
what runs inside the steps here is the sharded self-match job of polyfuzz_amd/pipeline.py with tests/cpu_engine.py in place of the
device -- so that the path from `python bench.py --gpus 2` to the job's collectives is exercised on a box without a GPU
(tests/test_bench_launch_cpu.py).  NOT a measurement and not a bench line: the record has no metric / value keys."""
import zlib

import numpy as np


class _NoDevice:
    """the context of --rehearse-cpu: nothing to synchronise, nothing to time"""
    def sync(self): pass
    def prof_enable(self, level): pass
    def prof_reset(self): pass
    def event_record(self, i): pass


def rehearse_cpu(world, args, timed_steps):
    """NOT a measurement and not a bench line (no metric / value keys): the rank logic of `--gpus N` -- launcher, gloo rendezvous,
    barrier + max-over-ranks around the timed steps, the sharded self-match job of polyfuzz_amd/pipeline.py with its collective
    `symmetric_ok` question and its exchanges -- on a box without a GPU, the device replaced by tests/cpu_engine.py."""
    from polyfuzz_amd import pipeline, synth
    from tests.cpu_engine import GlooComm, OracleEngine
    names = synth.company_names(400, seed=3)
    bounds = pipeline.balanced_bounds(names, world.size)
    b, e = bounds[world.rank]
    job = pipeline.TfidfMatchJob(None, names[b:e], names, top_n=3, comm=GlooComm(world.dist), self_match=True, shard_offset=b,
                                 rows_per_rank=max(y - x for x, y in bounds), engine=OracleEngine(world.torch))
    wall, result = timed_steps(world, _NoDevice(), job.step, args.steps, args.warmup)
    idx, val = result.download()
    idx, val = job.whole_result(idx, val, [y - x for x, y in bounds])
    if world.rank != 0:
        return None
    return {"rehearsal": "--rehearse-cpu: tests/cpu_engine.py in place of the device -- NOT a measurement", "world": world.size,
            "steps": args.steps, "rows": len(names), "result_is_full": bool(job.result_is_full),
            "idx_crc32": zlib.crc32(np.ascontiguousarray(idx, np.int32).tobytes()),
            "val_crc32": zlib.crc32(np.ascontiguousarray(val, np.float64).tobytes())}


