"""Drop-in boundary conformance, CPU only.  Both trees call their package `model`, so the reference's call
surface is read in a subprocess (tests/sig_dump.py) and compared with the drop-in's: every reference parameter list must be
an exact PREFIX of ours (name, kind, default) and anything we add must be keyword-only with a default -- so every call
the reference's sr.py / infer.py / sample.py can make binds identically.  Skipped where the reference tree is absent
(the GPU box)."""
import json
import os
import subprocess
import sys

import pytest
import torch

from helpers import ROOT

PKG = os.path.join(ROOT, 'image-super-resolution-via-iterative-refinement_amd')
REF = os.environ.get('SR3_REFERENCE', '/root/reference')
needs_ref = pytest.mark.skipif(not os.path.isfile(os.path.join(REF, 'model', 'networks.py')),
                               reason='reference tree not present')


def _dump(root):
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'sig_dump.py'), root], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-800:]
    return json.loads(r.stdout.decode().strip().splitlines()[-1])


@needs_ref
def test_public_signatures_match_reference():
    ref, ours = _dump(REF), _dump(PKG)
    problems = []

    def check(name, a, b):
        if b[:len(a)] != a:
            problems.append('%s: reference %s vs drop-in %s' % (name, a, b[:len(a)]))
        for extra in b[len(a):]:
            if extra[1] != 'KEYWORD_ONLY' or extra[2] is None:
                problems.append('%s: extension parameter %s must be keyword-only with a default' % (name, extra))
    n = 0
    for k, v in ref.items():
        assert k in ours, k
        if isinstance(v, dict):
            for meth, s in v.items():
                if meth not in ours[k]:
                    problems.append('%s.%s missing from the drop-in' % (k, meth))
                else:
                    check(k + '.' + meth, s, ours[k][meth])
                    n += 1
        else:
            check(k, v, ours[k])
            n += 1
    assert not problems, '\n'.join(problems)
    assert n >= 60          # DDPM (17 methods), both GaussianDiffusion classes, both UNets, factories


@needs_ref
@pytest.mark.parametrize('init_type', ['normal', 'kaiming', 'orthogonal'])
def test_init_weights_schemes_draw_the_reference_weights(init_type):
    """model/networks.py:14-77: same seed => same weights as the reference's `init_weights(netG, init_type)`."""
    code = ('import sys, logging; sys.path.insert(0, %r); logging.disable(50); import torch; import model.networks as N;'
            'from helpers_opt import opt; torch.manual_seed(5); net = N.define_G(opt()); torch.manual_seed(9);'
            'N.init_weights(net, init_type=%r, scale=0.3, std=0.05);'
            'torch.save({k: v for k, v in net.state_dict().items() if k.startswith("denoise_fn.")}, sys.argv[1])')
    import tempfile
    tmp = tempfile.mkdtemp()
    with open(os.path.join(tmp, 'helpers_opt.py'), 'w') as f:
        f.write('import sys\nsys.path.insert(0, %r)\nfrom helpers import opt_for\n'
                'def opt():\n    o = opt_for("sr3_tiny", phase="val", gpu=False)\n    return o\n' % os.path.join(ROOT, 'tests'))
    outs = {}
    for tag, root in (('ref', REF), ('ours', PKG)):
        out = os.path.join(tmp, tag + '.pth')
        env = dict(os.environ, PYTHONPATH=tmp)
        r = subprocess.run([sys.executable, '-c', code % (root, init_type), out], env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, timeout=300)
        assert r.returncode == 0, r.stderr.decode()[-800:]
        outs[tag] = torch.load(out)
    assert set(outs['ref']) == set(outs['ours'])
    for k, v in outs['ref'].items():
        assert torch.equal(v, outs['ours'][k]), k


def test_finetune_norm_raises_like_the_reference():
    """model/model.py:26-40: no parameter name contains 'transformer' => Adam gets an empty list => ValueError."""
    sys.path.insert(0, PKG)
    import model as Model
    from helpers import opt_for
    opt = opt_for('sr3_tiny', phase='train', gpu=False)
    opt['model']['finetune_norm'] = True
    with pytest.raises(ValueError, match='empty parameter list'):
        Model.create_model(opt)


@needs_ref
def test_reference_cpu_baseline_tool_runs(tmp_path):
    """tools/ref_baseline.py (bench.py's cpu_baseline leg, kind "reference"): imports the reference in its own process,
    loads the drop-in's weights by state-dict key, times p_sample and optimize_parameters."""
    sys.path.insert(0, PKG)
    sys.path.insert(0, ROOT)
    import bench
    import model.networks as networks
    opt = bench.config_opt('ddpm_128')
    opt['gpu_ids'] = None
    torch.manual_seed(0)
    netG = networks.define_G(opt)
    state = str(tmp_path / 'state.pth')
    torch.save({'sd': dict(netG.state_dict()), 'x': torch.randn(1, 3, 128, 128), 'cond': None, 't': 1007}, state)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'ref_baseline.py'), '--ref', REF, '--config', 'ddpm_128',
                        '--state', state, '--threads', '4', '--budget', '2', '--max-steps', '1', '--train-steps', '1',
                        '--train-batch', '1'], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-800:]
    rec = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert rec['kind'] == 'reference' and rec['value'] > 0 and rec['train']['value'] > 0, rec


@needs_ref
@pytest.mark.parametrize('name', ['sr3_16_128', 'sr3_64_512', 'ddpm_128'])
def test_bench_configs_are_the_reference_json_configs(name):
    """bench.py's CONFIGS / config_opt restate the `model` subtree of the reference's JSON files (comments stripped as
    core/logger.py:21-33 does): same network, same schedules, same image size -- so the bench line measures the
    configuration BASELINE.json names."""
    import re
    sys.path.insert(0, ROOT)
    import bench
    c = bench.CONFIGS[name]
    with open(os.path.join(REF, c['ref_json'])) as f:
        txt = '\n'.join(line.split('//')[0] for line in f)
    ref = json.loads(txt)['model']
    got = bench.config_opt(name)['model']
    assert got['which_model_G'] == ref['which_model_G']
    for k, v in ref['unet'].items():
        assert got['unet'].get(k, 32 if k == 'norm_groups' else None) == v, (k, v)
    assert set(got['unet']) <= set(ref['unet']) | {'norm_groups'}
    for ph in ('train', 'val'):
        for k in ('schedule', 'n_timestep', 'linear_start', 'linear_end'):
            assert got['beta_schedule'][ph][k] == ref['beta_schedule'][ph][k], (ph, k)
    assert got['diffusion']['image_size'] == ref['diffusion']['image_size'] == c['size']
    assert got['diffusion']['conditional'] == ref['diffusion']['conditional'] == c['conditional']
    full = json.loads(txt)
    assert abs(full['train']['optimizer']['lr'] - c['lr']) < 1e-12


def test_committed_profile_summaries_cover_the_kernels_bench_reports():
    """bench.py's `roofline.traffic` / `sq_counters` come from the committed rocprofv3 counter summaries of the round, looked up by
    the exact symbol of the dominant kernel: both plans the bench line reports (default: the two-workgroups-per-CU split Winograd
    kernel of round 6; `exact_fp32`: the 8-wave kernel's fp32-MFMA instantiation) must be in them, with sane values, or those
    fields silently turn null."""
    sys.path.insert(0, ROOT)
    import bench
    with open(os.path.join(ROOT, 'profiles', bench.PROFILE_ROUND + '_hbm_traffic.json')) as f:
        traffic = json.load(f)
    with open(os.path.join(ROOT, 'profiles', bench.PROFILE_ROUND + '_sq_counters.json')) as f:
        sq = json.load(f)
    for sym in ('sr3::k_conv3x3_wino2<0>', 'sr3::k_conv3x3_wino<0, false, false, false>',
                'sr3::k_gemm1x1_split<2, true, false, 1>', 'sr3::k_gemm1x1_split<1, false, true, 1>', 'sr3::k_gemm1x1_split<1, false, false, 2>',
                'sr3::k_conv_igemm<64, 64, 1, 0>',      # (the plain GEMM kernel: qkv, its stride-2 form, its 64 x 64 tile -- no im2col split launch
                                                        #  is left in the default plan since the last session of round 6; the fp32 plan keeps the im2col kernel)
                'sr3::k_conv3x3_wino<0, false, true, true>', 'sr3::k_attention_v2<2, 2, true>'):
        assert 5e6 < traffic[sym]['hbm_bytes_per_launch'] < 1e9, (sym, traffic.get(sym))
        assert 0.05 < sq[sym]['mfma_busy'] < 1.0, (sym, sq.get(sym))
    # the stats file the bench line's avg launch time must agree with
    with open(os.path.join(ROOT, 'profiles', bench.PROFILE_ROUND + '_bench_kernel_stats.csv')) as f:
        head = f.read(4000)
    assert 'k_conv3x3_wino2<0>' in head
    with open(os.path.join(ROOT, 'profiles', bench.PROFILE_ROUND + '_bench.json')) as f:
        line = json.loads(f.read().strip().splitlines()[-1])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert key in line, key
    r = line['roofline']
    assert r['bound'] == 'mfma' and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9 and r['traffic'] and r['unit'] == 'TFLOP/s'
    assert abs(line['value'] - 16 / (2000 * line['ms_per_step'] * 1e-3)) < 1e-9 * line['value'] + 1e-12
    assert line['cpu_baseline']['kind'] in ('port', 'reference') and line['cpu_baseline']['cores'] >= 1
