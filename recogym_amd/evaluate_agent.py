"""verify_agents — reference: recogym/evaluate_agent.py:732-750.

A/B-test a dict of agents on the same environment: every agent sees identical env draws (common
random numbers — here by construction: env draws are keyed by (seed, user id, t)), and the result
is the DataFrame [Agent, 0.025, 0.500, 0.975] of Beta CTR-posterior quantiles."""
from copy import deepcopy

import pandas as pd
from scipy.stats.distributions import beta

from .bench_agents import evaluate_counts


def verify_agents(env, number_of_users, agents):
    stat = {'Agent': [], '0.025': [], '0.500': [], '0.975': []}
    for agent_id in agents:
        successes, failures = evaluate_counts(deepcopy(env), agents[agent_id], number_of_users)
        stat['Agent'].append(agent_id)
        stat['0.025'].append(beta.ppf(0.025, successes + 1, failures + 1))
        stat['0.500'].append(beta.ppf(0.500, successes + 1, failures + 1))
        stat['0.975'].append(beta.ppf(0.975, successes + 1, failures + 1))
    return pd.DataFrame().from_dict(stat)
