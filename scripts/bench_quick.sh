for nb in 2 3; do
  FM_NBUF=$nb FM_BENCH_C3=0 FM_BENCH_C4=0 FM_BENCH_F32=0 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('NBUF=$nb', 'K1 kernel_ms %.4f'%d['roofline']['kernel_ms'], 'frac %.3f'%d['roofline']['frac'], 'ms/pair %.4f'%d['ms_per_image_pair'], 'K2 crm kernel_ms %.4f'%d['classic_ratio_match']['kernel_ms'], 'self2nn %.4f'%d['self_2nn']['kernel_ms'])"
done
