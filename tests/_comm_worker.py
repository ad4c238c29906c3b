"""One rank of the two-process communicator test (tests/test_comm.py): no torch; the RCCL unique id travels through a file."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "polars-bio_amd"), os.path.join(ROOT, "tests")]
from _util import random_side  # noqa: E402
from oracle import oracle as O  # noqa: E402
from polars_bio_amd import _engine, distributed as D  # noqa: E402


def main():
    rank, world, idf, outf = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    eng = _engine.Engine(rank % _engine.device_count())
    if rank == 0:
        uid = _engine.comm_unique_id()
        with open(idf + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(idf + ".tmp", idf)
    else:
        for _ in range(600):
            if os.path.exists(idf):
                break
            time.sleep(0.1)
        uid = open(idf, "rb").read()
    comm = _engine.Comm(eng, uid, rank, world)
    rng = np.random.default_rng(21)
    nc = 5
    probe = random_side(rng, 150000, nc, 1_500_000, 400)
    build = random_side(rng, 50000, nc, 1_500_000, 400)
    ep, eb = O.overlap_fast(O.Index(O.Side(*build), nc), O.Side(*probe), True)
    o = np.lexsort((eb, ep)); ep, eb = ep[o], eb[o]
    lp, pid, lb, bid, _ = D.shard_sides(probe, build, nc, rank, world)
    ptrs = []

    def up(a):
        p = eng.dev_alloc(max(4 * len(a), 16)); eng.h2d(p, np.ascontiguousarray(a, np.int32)); ptrs.append(p); return p
    ps = eng.dev_side(up(lp[0]), up(lp[1]), up(lp[2]), len(pid), up(pid))
    bs = eng.dev_side(up(lb[0]), up(lb[1]), up(lb[2]), len(bid), up(bid))
    opts = _engine.make_opts(True, nc)
    ix = eng.index_build_dev(bs, opts)
    total = len(ep)
    op, ob = eng.dev_alloc(4 * total + 16), eng.dev_alloc(4 * total + 16)
    nt, nl, fits = comm.overlap_allgather_dev(ix, ps, opts, 3, op, ob, total)
    hp, hb = np.empty(total, np.int32), np.empty(total, np.int32)
    eng.d2h(hp, op); eng.d2h(hb, ob)
    o = np.lexsort((hb, hp)); hp, hb = hp[o], hb[o]
    ok = int(fits and nt == total and (hp == ep).all() and (hb == eb).all())
    np.savez(outf, ok=ok, p=hp, b=hb)
    ix.close(); comm.close(); eng.close()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
