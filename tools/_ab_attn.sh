for r in 1 2 3; do for v in 1 21 24 18; do
python bench.py --quick --steps 30 --harness attn_variant=$v 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('variant $v', d['value'], d['kernel_ms_per_step'].get('attention'), d['kernel_ms_per_step'].get('gemm'), d['ms_per_step'])
"
done; done
