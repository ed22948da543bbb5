"""placeholder, replaced below"""
from .static_params import *  # noqa
env_args = {
    'num_products': 10, 'num_users': 100, 'random_seed': 0,
    'prob_leave_bandit': 0.01, 'prob_leave_organic': 0.01,
    'prob_bandit_to_organic': 0.05, 'prob_organic_to_bandit': 0.25,
    'normalize_beta': False, 'with_ps_all': False,
}
env_1_args = {**env_args, 'K': 5, 'sigma_omega_initial': 1, 'sigma_omega': 0.1,
              'number_of_flips': 0, 'sigma_mu_organic': 3, 'change_omega_for_bandits': False,
              'normalize_beta': False}
