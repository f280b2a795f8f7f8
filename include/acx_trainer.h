/*
 * acx_trainer.h -- C ABI of libacx_trainer.so: host utilities of the PPO trainer built on libacx (ac_solver/agents/).
 *
 * NOT part of the accelerated hot path and not part of libacx.so (round 6: they were in acx.h until ABI 500): two third-party
 * generators the reference's trainer draws from, restated so that they run off the interpreter lock beside a rollout.  Plain host
 * code, no HIP.  Every function returns 0 on success, ACXT_E_INVAL on a bad argument (acxt_last_error() has the message).
 */
#ifndef ACX_TRAINER_H
#define ACX_TRAINER_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ACXT_OK 0
#define ACXT_E_INVAL (-1)

/* The permutations  np.random.seed(seed); for e in range(epochs): np.random.shuffle(b_inds)  leaves in b_inds = np.arange(n) -- what
 * the update loop of agents/training.py:121, 273-275 draws its minibatches from -- written to h_out [epochs, n].  NumPy's legacy
 * MT19937 / shuffle algorithm restated (a third-party dependency of the reference, pinned against numpy in tests/test_agents_cpu.py);
 * runs without the interpreter lock, so the driver computes it on a thread beside the rollout.  n < 2^32. */
int acxt_np_shuffle_epochs(uint32_t seed, int64_t n, int epochs, int64_t *h_out);

/* The draws of n consecutive curriculum decisions (training.py:199-221 of the reference, after its first round:
 * `len(solved) == 0 or (unsolved and random.uniform(0, 1) > repeat_solved_prob)` -> random.choice(list(unsolved)), else
 * random.choice(list(solved))) for FIXED sizes of the two lists.  mt_state [624] / *mt_pos: the state of Python's global `random`
 * generator as random.getstate()[1] holds it, advanced in place exactly as n calls of the Python code advance it (CPython 3.10's
 * Random restated).  which[i] = 0: element index[i] of the unsolved list, 1: of the solved list. */
int acxt_py_curriculum_draws(uint32_t *mt_state, int32_t *mt_pos, int64_t n, int64_t n_solved, int64_t n_unsolved,
                             double repeat_solved_prob, uint8_t *which, int64_t *index);

const char *acxt_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
