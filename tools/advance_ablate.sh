#!/bin/bash
# one step with everybody in bandit state is not reachable; use steps 30..40 of a 4M-user run
for ab in 0 64 128 256 448; do echo -n "ablate=$ab  "; RECOGYM_ABLATE=$ab python tools/advance_probe.py 10000 20 4000000 40 | grep ouc | sed 's/.*advance/advance/'; done
