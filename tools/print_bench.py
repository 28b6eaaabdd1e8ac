"""One-line digest of a bench.py JSON line: python tools/print_bench.py <file>"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(f"{d['value']} {d['unit']}, {d['ms_per_step']} ms/step; {r['kernel']}: {r['achieved']} TF = {r['frac']} in the timed region, "
      f"{r.get('isolated', {}).get('achieved')} = {r.get('isolated', {}).get('frac')} isolated; batched "
      f"{d.get('batched', {}).get('value_this_rank')}; bit-exact {d.get('accuracy', {}).get('logits_bit_exact')}; "
      f"hbm {[(k.split()[0] + (' x4' if 'x4' in k else ''), v['frac_of_hbm_peak']) for k, v in d.get('hbm_bound_layers', {}).items() if isinstance(v, dict)]}")
