#!/bin/bash
# headline step with different K1 timing-event densities (FM_ASYNC_TIME_EVERY)
for n in 4 1 10 4 1 10; do
  FM_ASYNC_TIME_EVERY=$n python bench.py --no-legs --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('TIME_EVERY=$n ms/pair %.4f value %.4e K1 %s launches_timed %s' % (d['ms_per_image_pair'], d['value'], d['roofline']['kernel_ms'], d['roofline']['kernel_launches_timed']))"
done
