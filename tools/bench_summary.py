#!/usr/bin/env python
"""tools/bench_summary.py <bench JSON file> -- one line with the rates of a default `bench.py` line (headline, step forms,
dense / developed state, app figure, app run per quarter, fast mode), for same-box comparisons of library builds."""
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ar=d.get('app_run',{})
print(sys.argv[1], 'headline %.0f (%.4f ms)' % (d['value'], d['ms_per_step']), 'forms', d['config'].get('step_form',{}).get('one_kernel_steps'), 'dense %.0f' % d['dense_state']['value'], 'developed %.0f' % d['developed_state']['value'], 'app_fig %.0f' % d['app_figure']['value'], 'app_run %.0f' % ar.get('value',0), [round(q['value']) for q in ar.get('quarters',[])], 'fast bubble %.0f dense %.0f' % (d['fast_math']['bubble']['value'], d['fast_math']['dense']['value']))
