"""The randomised parity sweep of tests/test_gpu_fuzz.py over many cases, spread over worker processes that share the GPU
(a case is a few milliseconds of GPU work and tens of milliseconds of host work -- the reference build, the plain-C
oracle for the rare arbitration -- so one process leaves the GPU idle).

usage: python scripts/fuzz_sweep.py <n_cases> [--workers 8] [--first 0] [--runs 1]
Prints every failing case, every escape-hatch exit (the test prints those itself) and one JSON line with the tally per
run; exit status 1 if any case failed or the exits exceed the caps of test_fuzz_escape_hatches_stay_rare."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "gaussian-pcloud-render_amd")]


def worker(k, n_workers, first, n_cases, q):
    import torch
    import test_gpu_fuzz as F
    dev = torch.device("cuda:0")
    failed = []
    for i in range(first + k, first + n_cases, n_workers):
        try:
            F.test_random_case_matches_reference_build(i, dev)
        except AssertionError as e:
            failed.append((i, str(e).splitlines()[0][:300]))
            print("FAILED case %d: %s" % (i, failed[-1][1]), flush=True)
    q.put((dict(F.TALLY), failed))


def main():
    import multiprocessing as mp
    a = sys.argv[1:]
    n_cases = int(a[0])
    opt = dict(workers=8, first=0, runs=1)
    for j in range(1, len(a), 2):
        opt[a[j].lstrip("-")] = int(a[j + 1])
    ctx = mp.get_context("spawn")
    bad = False
    for run in range(opt["runs"]):
        t0 = time.time()
        q = ctx.Queue()
        ps = [ctx.Process(target=worker, args=(k, opt["workers"], opt["first"], n_cases, q)) for k in range(opt["workers"])]
        for p in ps:
            p.start()
        res = [q.get() for _ in ps]
        for p in ps:
            p.join()
        tally, failed = {}, []
        for t, f in res:
            for key, v in t.items():
                tally[key] = tally.get(key, 0) + v
            failed += f
        cases = tally["cases_exit_4x_reference"] + tally["cases_exit_conditioning"]
        rows = tally["rows_exit_4x_reference"] + tally["rows_exit_conditioning"]
        import test_gpu_fuzz as F
        over = cases > max(1, int(F.MAX_CASE_FRACTION * n_cases)) or rows > max(2, int(F.MAX_ROW_FRACTION * tally["rows"]))
        print(json.dumps(dict(run=run, n_cases=n_cases, first=opt["first"], seconds=round(time.time() - t0, 1), tally=tally,
                              failed=sorted(failed), exits_over_cap=over)), flush=True)
        bad = bad or over or bool(failed)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
