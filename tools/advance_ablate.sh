#!/bin/bash
for ab in 0 32 64 96 128 256 384; do echo "ablate=$ab"; RECOGYM_ABLATE=$ab python tools/advance_probe.py 10000 20 4000000 40 | head -1; done
