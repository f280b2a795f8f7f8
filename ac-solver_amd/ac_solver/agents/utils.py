"""Helpers of the PPO trainer (reference: ac_solver/agents/utils.py:10-34)."""
from ast import literal_eval


def load_initial_states_from_text_file(states_type):
    """The Miller-Schupp presentations as lists, ordered by hardness: "solved" -> the greedy-solved ones,
    "all" -> those followed by the rest (ac_solver/search/miller_schupp/data/*.txt; the files are produced
    on first use by this build's own searches, see ac_solver.search.miller_schupp.data_files)."""
    assert states_type in ["solved", "all"], "states_type must be 'solved' or 'all'"
    from ac_solver.search.miller_schupp.data_files import ensure_data_file

    file_name = f"{'greedy_solved' if states_type == 'solved' else 'all'}_presentations.txt"
    with open(ensure_data_file(file_name)) as f:
        initial_states = [literal_eval(line.strip()) for line in f if line.strip()]
    print(f"Loaded {len(initial_states)} presentations from {file_name}.")
    return initial_states
