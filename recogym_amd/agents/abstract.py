"""The policy interface the env calls — reference: recogym/agents/abstract.py:9-34.

`act(observation, reward, done) -> {'t','u','a','ps','ps-a'}`, `train(...)`, `reset()`.
Agents that can run ON THE DEVICE inside the batched step loop additionally expose
`device_policy() -> dict(policy=, policy_seed=, ouc=)`; `env.generate_logs`, `test_agent` and
`verify_agents` then never call `act` — the same draws are made by the HIP kernels.  Their
Python `act` is the host form of the identical policy (same counter-RNG draws), used by the
per-user gym path (`env.step_offline`), so both routes log the same actions.
"""


class Agent:
    def __init__(self, config):
        self.config = config

    def act(self, observation, reward, done):
        return {'t': observation.context().time(), 'u': observation.context().user()}

    def train(self, observation, action, reward, done=False):
        pass

    def reset(self):
        pass
