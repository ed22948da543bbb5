# Round 4, GPU call 13: the host-side additions on the box — weight_history_function through the per-user gym path, the env drop-in
# and feature-feed tests, test_agent with the pickle cache on the device path.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_env_dropin.py tests/test_feature_feed.py -m gpu -q > $O/gpu_tests13.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests13.log; tail -5 $O/gpu_tests13.log | cut -c1-400
RECOGYM_CACHE_DIR=/tmp/rgcache python - <<'PY'
import time, recogym_amd as recogym
from recogym_amd import Configuration, env_1_args
from recogym_amd.agents import LogregMulticlassIpsAgent, logreg_multiclass_ips_args
env = recogym.make('reco-gym-v1'); env.init_gym({**env_1_args, 'random_seed': 42, 'num_products': 20, 'K': 6})
agent = LogregMulticlassIpsAgent(Configuration({**logreg_multiclass_ips_args, 'num_products': 20, 'random_seed': 7}))
for rep in range(2):
    t0 = time.time(); q = recogym.test_agent(env, agent, 400, 20000, with_cache=True); print('test_agent with_cache run', rep, [round(x, 5) for x in q], round(time.time() - t0, 2), 's')
PY
ls /tmp/rgcache
