# in-model A/B of the attention forms (EXPERIMENTS.md R5.9): 36 = the 8-wave kernel at every size, 1 = the shipped policy (persistent
# form at this batch), 24 = persistent with 2 loader waves.  Prints img/s, attention ms/step, GEMM ms/step, ms/step per run.
for r in 1 2 3; do for v in 36 1 24; do
python bench.py --quick --steps 30 --harness attn_variant=$v 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('variant $v', d['value'], d['kernel_ms_per_step'].get('attention'), d['kernel_ms_per_step'].get('gemm'), d['ms_per_step'])
"
done; done
