"""On-disk data of the Miller-Schupp experiments (reference: ac_solver/search/miller_schupp/data/*.txt and the
`_solved` / `_unsolved` / `_paths` outputs of trivialize_miller_schupp_through_search, miller_schupp.py:160-175).

One Python literal per line:
  all_presentations.txt            1190 presentations of n = 1..7, max_w_len = 7: the greedy-solved ones first
                                   (generator order), then the rest -- "sorted by hardness" for the PPO curriculum
  greedy_solved_presentations.txt  the 533 presentations greedy_search trivialises with a 1e6-node budget
  greedy_search_paths.txt          their paths, aligned line by line, in the LEGACY encoding of the published file:
                                   actions are 1-based and the root entry is (0, length) instead of (-1, length)
  bfs_solved_presentations.txt     the 278 presentations bfs trivialises with a 1e6-node budget (cyclic reduction on)

The files are not shipped: `ensure_data_file` produces them with this build's own searches on the GPU
(well under a second of device time) the first time one is asked for, into ac_solver/search/miller_schupp/data/.
First use under N ranks (torchrun starts N trainers on a fresh box): ONE process generates, holding an exclusive
`flock` on `<dir>/.lock`; the others block on the lock and find the files when they get it.  Every file is written
under a temporary name and renamed into place (`os.replace`), so a reader never sees a half-written file.
"""
import contextlib
import fcntl
import os
import tempfile
from ast import literal_eval

import numpy as np

DATA_DIR = os.path.join(os.path.dirname(os.path.realpath(__file__)), "data")
FILES = ("all_presentations.txt", "greedy_solved_presentations.txt", "greedy_search_paths.txt", "bfs_solved_presentations.txt")


def to_legacy_path(path):
    """[(-1, l0), (a, l), ...] -> [(0, l0), (a + 1, l), ...] (the encoding of the published greedy_search_paths.txt)"""
    return [(int(a) + 1, int(l)) for a, l in path]


def from_legacy_path(path):
    return [(int(a) - 1, int(l)) for a, l in path]


def write_literals(rows, filepath):
    """One literal per line, atomically: the text goes to a temporary file of the same directory which is then renamed over
    `filepath` -- a concurrent reader sees the old file or the whole new one."""
    d = os.path.dirname(os.path.abspath(filepath))
    fd, tmp = tempfile.mkstemp(prefix="." + os.path.basename(filepath) + ".", suffix=".tmp", dir=d)
    try:
        with os.fdopen(fd, "w") as f:
            for row in rows:
                f.write(f"{row}\n")
            f.flush()
            os.fsync(f.fileno())
        os.chmod(tmp, 0o644)
        os.replace(tmp, filepath)
    except BaseException:
        with contextlib.suppress(OSError):
            os.unlink(tmp)
        raise


@contextlib.contextmanager
def _dir_lock(d):
    """Exclusive advisory lock on `<d>/.lock` (flock: released by the kernel when the holder dies)."""
    os.makedirs(d, exist_ok=True)
    fd = os.open(os.path.join(d, ".lock"), os.O_RDWR | os.O_CREAT, 0o644)
    try:
        fcntl.flock(fd, fcntl.LOCK_EX)
        yield
    finally:
        with contextlib.suppress(OSError):
            fcntl.flock(fd, fcntl.LOCK_UN)
        os.close(fd)


def read_literals(filepath):
    with open(filepath) as f:
        return [literal_eval(line.strip()) for line in f if line.strip()]


def replay_paths(presentations, paths, cyclical=False, want_final=False):
    """Apply the actions of many search paths to their presentations in ONE launch (acx_replay_paths: a lane per path); returns,
    per path, the list of total lengths after each move -- and the presentations the paths end at with `want_final`.
    `presentations`: rows of one max_relator_length; `paths`: search paths [(-1, l0), (a, l), ...] (the root entry is skipped).
    A path is valid when its list equals its recorded lengths and ends at 2 (breadth_first.py:113-126).  A move on which the
    reference's ACMove raises does so here as well (AssertionError / IndexError, as envs.ac_moves.ACMove)."""
    import ctypes as C

    from ac_solver import _acx

    _acx.require_device()
    rows = _acx.as_i8_rows(np.asarray(presentations, dtype=np.int8).reshape(len(paths), -1))
    n, L = rows.shape[0], rows.shape[1] // 2
    acts = [np.array([int(a) for a, _ in p[1:]], np.int32) for p in paths]
    offsets = np.zeros(n + 1, np.int64)
    offsets[1:] = np.cumsum([len(a) for a in acts])
    flat = np.ascontiguousarray(np.concatenate(acts) if n and offsets[-1] else np.zeros(0, np.int32), np.int32)
    out = np.zeros(max(int(offsets[-1]), 1), np.int32)
    err = np.zeros(max(n, 1), np.uint8)
    final = np.zeros_like(rows) if want_final else None
    _acx.check(_acx.lib.acx_replay_paths(_acx.ptr(rows, C.c_int8), n, L, int(bool(cyclical)), _acx.ptr(flat, C.c_int32) if len(flat) else None,
                                         _acx.ptr(offsets, C.c_int64), _acx.ptr(out, C.c_int32), _acx.ptr(err, C.c_uint8),
                                         None if final is None else _acx.ptr(final, C.c_int8)), "acx_replay_paths")
    for i in np.flatnonzero(err[:n]):
        raise (IndexError if err[i] == _acx.ERR_INDEX else AssertionError)(f"path {i}: a move of the path is not valid on {rows[i].tolist()} (code {int(err[i])})")
    lens = [out[offsets[i]:offsets[i + 1]].tolist() for i in range(n)]
    return (lens, final) if want_final else lens


def replay_path(presentation, path, cyclical=False):
    """Apply the actions of a search path to `presentation` on the GPU; returns the list of total lengths after each
    move.  A path is valid when this equals its recorded lengths and ends at 2 (breadth_first.py:113-126)."""
    return replay_paths([presentation], [path], cyclical)[0]


def make_data_files(max_nodes_to_explore=10**6, out_dir=None, verbose=True):
    """Run greedy_search and bfs over the 1190 Miller-Schupp presentations and write the four files."""
    from ac_solver import _acx
    from ac_solver.search._common import run_search_groups
    from ac_solver.search.miller_schupp.miller_schupp import generate_miller_schupp_presentations

    out_dir = out_dir or DATA_DIR
    os.makedirs(out_dir, exist_ok=True)
    solved, unsolved, paths, bfs_solved = [], [], [], []
    all_rows = []
    for n in range(1, 8):
        by_len = generate_miller_schupp_presentations(n, 7)
        all_rows.append([p for lenw in range(1, 8) for p in by_len.get(lenw, [])])
    arrs = [np.array(rows, dtype=np.int8) for rows in all_rows]
    greedy_all = run_search_groups(_acx.SEARCH_GREEDY, arrs, max_nodes_to_explore, False)  # all seven widths in one call (acx_search_groups)
    bfs_all = run_search_groups(_acx.SEARCH_BFS, arrs, max_nodes_to_explore, True)
    for n in range(1, 8):
        rows, greedy, bfs = all_rows[n - 1], greedy_all[n - 1], bfs_all[n - 1]
        for p, (ok, path, _), (bok, _, _) in zip(rows, greedy, bfs):
            if ok:
                solved.append(p)
                paths.append(to_legacy_path(path))
            else:
                unsolved.append(p)
            if bok:
                bfs_solved.append(p)
        if verbose:
            print(f"Miller-Schupp n = {n}: {len(rows)} presentations, {len(solved)} greedy-solved and {len(bfs_solved)} bfs-solved so far", flush=True)
    write_literals(solved + unsolved, os.path.join(out_dir, "all_presentations.txt"))
    write_literals(solved, os.path.join(out_dir, "greedy_solved_presentations.txt"))
    write_literals(paths, os.path.join(out_dir, "greedy_search_paths.txt"))
    # published order: that of all_presentations.txt
    order = {tuple(p): k for k, p in enumerate(solved + unsolved)}
    write_literals(sorted(bfs_solved, key=lambda p: order[tuple(p)]), os.path.join(out_dir, "bfs_solved_presentations.txt"))
    return out_dir


def _have_all(d):
    return all(os.path.exists(os.path.join(d, name)) for name in FILES)


def ensure_data_files(data_dir=None, generate=None):
    """All four files present in `data_dir` (default: the package's data directory), produced at most ONCE however many
    processes ask at the same time: whoever gets the directory lock first generates (files appear by rename), the others wait
    on the lock and then find them.  `generate(out_dir)` defaults to `make_data_files` (this build's searches on the GPU)."""
    d = data_dir or DATA_DIR
    if _have_all(d):
        return d
    with _dir_lock(d):
        if not _have_all(d):
            print(f"Miller-Schupp data files not found: running the searches once to produce {d} ...", flush=True)
            (generate or (lambda out: make_data_files(out_dir=out)))(d)
    return d


def ensure_data_file(name, data_dir=None, generate=None):
    assert name in FILES, f"unknown data file {name}"
    return os.path.join(ensure_data_files(data_dir, generate), name)


if __name__ == "__main__":
    print("wrote", make_data_files())
