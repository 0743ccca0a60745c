import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])['survey_config5']
for tag in ('one_by_one', 'two_sources_of_a_frequency_together'):
    b = d[tag]
    print(tag, 'best ms/source %.0f' % b['ms_per_source'], 'first-repeat ms/source %.0f' % b['first_repeat']['ms_per_source'], 'cycles', b['cycles'])
    for pf in b['per_frequency']:
        for i, c in enumerate(pf['calls']):
            print('  f=%.2f call %d: total %.2f s = sources %.2f + hierarchy %.2f + solve %.2f; reserved before %.1f after %.1f GB; cycle wall ms %s' % (
                pf['frequency'], i, c['seconds'], c['source_vectors_s'], c['hierarchy_s'], c['solve_s'], c['reserved_gb_before'], c['reserved_gb_after'], c['cycle_wall_ms']))
