"""PPO on the device-resident AC environments -- counterpart of the reference's ac_solver/agents package
(SURVEY.md section 8(f)-1): same module names, functions and command-line flags; the rollout runs on
`ACVecEnv` (one HIP kernel launch per step, observations / rewards / flags written straight into the
rollout buffers) instead of a Python loop over gymnasium environments."""
