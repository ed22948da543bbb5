from .abstract import Agent
from .random_agent import RandomAgent, random_args
from .organic_user_count import OrganicUserEventCounterAgent, organic_user_count_args
from .last_view_table import LastViewTableAgent
from .logreg_frozen import LogregFrozenAgent
from .logreg_ips import LogregMulticlassIpsAgent, logreg_multiclass_ips_args
from .feature_feed import train_data_from_log
from .bandit_mf import BanditMFSquareAgent, bandit_mf_square_args
