import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ar=d.get('app_run',{})
print(sys.argv[1], 'headline %.0f (%.4f ms)' % (d['value'], d['ms_per_step']), 'forms', d['config'].get('step_form',{}).get('one_kernel_steps'), 'dense %.0f' % d['dense_state']['value'], 'developed %.0f' % d['developed_state']['value'], 'app_fig %.0f' % d['app_figure']['value'], 'app_run %.0f' % ar.get('value',0), [round(q['value']) for q in ar.get('quarters',[])], 'fast bubble %.0f dense %.0f' % (d['fast_math']['bubble']['value'], d['fast_math']['dense']['value']))
