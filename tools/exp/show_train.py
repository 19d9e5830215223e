"""print the one-line summaries of train_bench JSON files"""
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, '\n  ', round(d['iters_per_s'], 1), 'it/s psnr', round(d['psnr_start'], 2), '->', round(d['psnr_end'], 2), 'N',
              d['num_gaussians_start'], '->', d['num_gaussians_end'], d.get('phase_ms_median'))
        print('  ', [n for s, n in d['refinements']][::4])
    except Exception as e:
        print(f, 'ERR', e)
