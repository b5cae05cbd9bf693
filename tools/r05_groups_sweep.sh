#!/bin/bash
# ON THE GPU BOX: strata groups (chunks per 8x8 pixel block) x frames in flight, ms per frame of a 1/N shard of the C2 frame
for g in ${GROUPS_LIST:-1 2 4 8}; do echo "groups $g"; JTX_STRATA_GROUPS=$g timeout -k 10 200 python3 tools/tools_shard_time.py ${WL:-cornell_1920x1080_64spp_d8} ${FRAMES:-20} 2>&1 | grep "world"; done
