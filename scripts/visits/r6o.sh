#!/bin/bash
# round 6 visit o: where the step stands -- kernel trace of the headline step + the four other BASELINE configurations
bash scripts/archive/gpu_r2.sh r6o profbf configs 2>&1 | cut -c1-2600
