"""First use of the Miller-Schupp data files under N processes (torchrun starts N trainers on a fresh box; the files are
git-ignored): ONE process generates, nobody ever reads a half-written file.  Reference: the files are shipped with the
package there (ac_solver/search/miller_schupp/data/*.txt, agents/utils.py:28); here they are produced on first use."""
import multiprocessing as mp
import os
import time

from ac_solver.search.miller_schupp import data_files as DF

ROWS = [[1, 0, 2, 0]] * 2000


def _slow_generate(out_dir):
    """stands in for make_data_files (which needs a GPU): logs the call, writes slowly, file by file"""
    with open(os.path.join(out_dir, "calls.log"), "a") as f:
        f.write(f"{os.getpid()}\n")
    for name in DF.FILES:
        time.sleep(0.15)
        DF.write_literals(ROWS, os.path.join(out_dir, name))


def _racer(d, barrier, q):
    barrier.wait()
    try:
        path = DF.ensure_data_file("all_presentations.txt", data_dir=d, generate=_slow_generate)
        rows = DF.read_literals(path)
        others = [len(DF.read_literals(os.path.join(d, n))) for n in DF.FILES]
        q.put((len(rows), others))
    except BaseException as e:  # noqa: BLE001
        q.put(repr(e))


def test_racing_processes_generate_once_and_read_whole_files(tmp_path):
    d = str(tmp_path / "data")
    ctx = mp.get_context("fork")
    n = 4
    barrier, q = ctx.Barrier(n), ctx.Queue()
    procs = [ctx.Process(target=_racer, args=(d, barrier, q)) for _ in range(n)]
    for p in procs:
        p.start()
    got = [q.get(timeout=60) for _ in procs]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert got == [(len(ROWS), [len(ROWS)] * 4)] * n, got
    with open(os.path.join(d, "calls.log")) as f:
        assert len(f.read().split()) == 1, "more than one process generated the files"
    assert not [x for x in os.listdir(d) if x.endswith(".tmp")]


def test_write_is_atomic_for_a_concurrent_reader(tmp_path):
    """a reader polling the path while it is rewritten sees the old file or the whole new one, never a prefix"""
    path = str(tmp_path / "f.txt")
    DF.write_literals([[0]] * 10, path)
    ctx = mp.get_context("fork")
    stop = ctx.Event()

    def writer():
        k = 0
        while not stop.is_set():
            DF.write_literals([[k]] * (5000 + k % 7), path)
            k += 1

    p = ctx.Process(target=writer)
    p.start()
    try:
        t0 = time.time()
        while time.time() - t0 < 1.5:
            rows = DF.read_literals(path)
            assert len(rows) == 10 or (len(rows) >= 5000 and len({tuple(r) for r in rows}) == 1 and len(rows) == 5000 + rows[0][0] % 7)
    finally:
        stop.set()
        p.join(10)


def test_failed_generation_leaves_no_partial_file_and_releases_the_lock(tmp_path):
    d = str(tmp_path / "data")

    def broken(out_dir):
        DF.write_literals(ROWS, os.path.join(out_dir, DF.FILES[0]))
        raise RuntimeError("no device")

    try:
        DF.ensure_data_files(d, generate=broken)
        raise AssertionError("expected the generator's error")
    except RuntimeError:
        pass
    assert not [x for x in os.listdir(d) if x.endswith(".tmp")]
    DF.ensure_data_files(d, generate=_slow_generate)  # the lock is free again and the next caller completes the set
    assert DF._have_all(d)


def test_importing_the_data_package_does_no_gpu_work():
    import ac_solver.search.miller_schupp.data as data

    assert callable(data.ensure) and data.FILES == DF.FILES
