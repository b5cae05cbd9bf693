#!/bin/bash
# local, after tools/r06_collect.sh + the four bench lines ran on the GPU box (their outputs merged into gpurun_out/): copy the records into profiles/
cd "$(dirname "$0")/.." || exit 1
G=gpurun_out
cp $G/r06_pmc_merged.json profiles/r06_pmc.json
for f in r06_bench r06_bench_c3_atrium r06_bench_c5_mixed r06_bench_c1; do tail -1 $G/$f.json > profiles/$f.json; done
cp $G/ks_r06_serial_cornell_1920x1080_64spp_d8.csv profiles/r06_kernel_stats.csv
cp $G/ks_r06_serial_atrium_1920x1080_64spp_d8.csv profiles/r06_kernel_stats_c3_atrium.csv
cp $G/ks_r06_serial_mixed_1920x1080_128spp_d8.csv profiles/r06_kernel_stats_c5_mixed.csv
cp $G/ks_r06_inflight_cornell_1920x1080_64spp_d8.csv profiles/r06_kernel_stats_in_flight.csv
cp $G/ks_r06_inflight_atrium_1920x1080_64spp_d8.csv profiles/r06_kernel_stats_in_flight_c3_atrium.csv
cp $G/ks_r06_inflight_mixed_1920x1080_128spp_d8.csv profiles/r06_kernel_stats_in_flight_c5_mixed.csv
cp $G/ks_r06_progressive_1.csv profiles/r06_kernel_stats_progressive_spp1.csv
cp $G/ks_r06_progressive_8.csv profiles/r06_kernel_stats_progressive_spp8.csv
python3 - <<'PY'
import json, csv
for f in ('r06_bench', 'r06_bench_c3_atrium', 'r06_bench_c5_mixed', 'r06_bench_c1'):
    d = json.loads(open('profiles/%s.json' % f).read()); r = d['roofline']; im = r.get('issue_model') or {}
    print(f, 'ms_per_step', d['ms_per_step'], 'value', d['value'], 'frac', r['frac'], '/', r['lone']['frac'], 'useful', r['useful_frac'], '/', r['lone']['useful_frac'],
          'attainable', r.get('useful_frac_attainable'), 'lone ms', r['lone']['kernel_ms'], 'stale', r['pmc_stale'], (im.get('calibration') or {}).get('stale'),
          'co_issue', im.get('co_issue', {}).get('issue_slots_busy'), im.get('co_issue', {}).get('half_rate_pipe_busy'), 'busy', im.get('busy'), im.get('busy_calibrated'),
          'device', d.get('ms_per_step_device'), 'blocking', d.get('ms_per_step_host_blocking'))
for f in ('r06_kernel_stats', 'r06_kernel_stats_c3_atrium', 'r06_kernel_stats_c5_mixed', 'r06_kernel_stats_progressive_spp1', 'r06_kernel_stats_progressive_spp8'):
    for r in csv.DictReader(open('profiles/%s.csv' % f)):
        if 'k_render_paths' in r['Name'] or 'k_resolve_prog' in r['Name']:
            print(f, r['Name'][:48], r['Calls'], round(float(r['AverageNs']) / 1e6, 3))
PY
