"""The reference's OWN scripts -- infer.py, sr.py, sample.py, executed as files through runpy -- on top of the drop-in
`model` / `data` / `core.metrics` packages, laid out as INTEGRATION.md section 1 describes (tests/run_reference_script.py).
SURVEY.md 4 ("API-level") and 8b: the boundary is the Python surface those scripts call, so this is the test that the
scripts really run unchanged: argparse -> core/logger.py parse -> create_dataset / create_dataloader -> create_model ->
set_new_noise_schedule -> the train / validation / sample loops -> metrics, image files, checkpoints.

This container has no GPU and the engine has no CPU fallback, so there are two kinds of run:
  * as is (options forced to gpu_ids = None, the reference's CPU switch): the script must get through everything that is
    not arithmetic and stop with Sr3Error at its FIRST engine call, at the script line where that call is made;
  * with the engine calls replaced by CPU stand-ins (train step, Adam, reverse loop, uint8 transform, metrics -- the
    oracle's restatements): the script must run to its last line and leave the reference's files behind.
The reference tree only exists in the build container: skipped elsewhere (the GPU box)."""
import json
import os
import re
import subprocess
import sys

import pytest

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REF, 'infer.py')), reason='reference tree not present')


def _config(tmp_path, name, **train):
    """The reference's JSON with the PNG triplets it ships (datatype img: lmdb is not installed) and a short run."""
    text = open(os.path.join(REF, 'config', name)).read()
    text = text.replace('"datatype": "lmdb"', '"datatype": "img"').replace('dataset/ffhq_16_128', 'dataset/celebahq_16_128')
    text = re.sub(r'"n_iter": \d+', '"n_iter": %d' % train.get('n_iter', 4), text)
    path = tmp_path / name
    path.write_text(text)
    return str(path)


def _run(tmp_path, script, args, *flags):
    work = tmp_path / 'tree'
    work.mkdir(exist_ok=True)
    env = dict(os.environ, SR3_DP='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(HERE, 'run_reference_script.py'), *flags, REF, str(work), script] + args,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900, env=env)
    return p.returncode, p.stdout, work


def _script_line(script, needle):
    for i, line in enumerate(open(os.path.join(REF, script)), 1):
        if needle in line:
            return i
    raise AssertionError(needle)


def _exp_dir(work):
    exps = sorted((work / 'experiments').iterdir())
    assert len(exps) == 1
    return exps[0]


def test_infer_py_stops_at_its_first_engine_call(tmp_path):
    cfg = _config(tmp_path, 'sr_sr3_16_128.json')
    rc, out, work = _run(tmp_path, 'infer.py', ['-c', cfg, '-debug'], '--force-cpu')
    assert rc == 3, out[-3000:]
    # everything before the validation loop ran through the reference's own file ...
    for msg in ('Initial Dataset Finished', 'Model [DDPM] is created.', 'Initial Model Finished', 'Begin Model Inference.'):
        assert msg in out, msg
    assert 'Dataset [LRHRDataset - CelebaHQ] is created.' in out
    # ... and the first engine call (the uint8 -> fp32 batch transform behind the loader) refuses to run without a GPU
    assert 'SR3ERROR:' in out and 'no CPU fallback' in out
    assert 'SCRIPT_LINE: %d' % _script_line('infer.py', 'enumerate(val_loader)') in out


def test_sr_py_val_stops_at_its_first_engine_call(tmp_path):
    cfg = _config(tmp_path, 'sr_sr3_16_128.json')
    rc, out, _ = _run(tmp_path, 'sr.py', ['-p', 'val', '-c', cfg, '-debug'], '--force-cpu')
    assert rc == 3, out[-3000:]
    assert 'Begin Model Evaluation.' in out and 'SR3ERROR:' in out


def test_infer_py_runs_to_its_end_on_standins(tmp_path):
    cfg = _config(tmp_path, 'sr_sr3_16_128.json')
    rc, out, work = _run(tmp_path, 'infer.py', ['-c', cfg, '-debug'], '--force-cpu', '--standins')
    assert rc == 0, out[-3000:]
    assert "CALL: ('reverse_loop', (1, 3, 128, 128), 10, True)" in out          # -debug: 10 steps, infer.py:70 continous=True
    res = _exp_dir(work) / 'results'
    assert sorted(p.name for p in res.iterdir()) == ['0_1_hr.png', '0_1_inf.png', '0_1_sr.png', '0_1_sr_process.png']
    from PIL import Image
    assert Image.open(res / '0_1_sr.png').size == (128, 128)
    # the grid of the 11 + 1 frames (infer.py:84-86; make_grid nrow = floor(sqrt(12)) = 3, padding 2)
    assert Image.open(res / '0_1_sr_process.png').size == (3 * 130 + 2, 4 * 130 + 2)


def test_sr_py_train_runs_to_its_end_on_standins(tmp_path):
    cfg = _config(tmp_path, 'sr_sr3_16_128.json', n_iter=4)
    rc, out, work = _run(tmp_path, 'sr.py', ['-p', 'train', '-c', cfg, '-debug'], '--force-cpu', '--standins')
    assert rc == 0, out[-3000:]
    assert out.count("CALL: ('train_step', (1, 3, 128, 128))") == 4 and out.count("CALL: ('adam',)") == 4
    # -debug: print_freq 2, val_freq 2, save_checkpoint_freq 3 (core/logger.py:62-71)
    assert '<epoch:  2, iter:       2> l_pix:' in out and '<epoch:  4, iter:       4> l_pix:' in out
    assert out.count('# Validation # PSNR:') == 2
    assert out.count("CALL: ('reverse_loop', (1, 3, 128, 128), 10, False)") == 2      # sr.py:115 continous=False
    assert "('add_scalar', 'psnr'," in out and "('add_scalar', 'l_pix'," in out
    exp = _exp_dir(work)
    ck = sorted(p.name for p in (exp / 'checkpoint').iterdir())
    assert ck == ['I3_E3_gen.pth', 'I3_E3_opt.pth'], ck
    import torch
    gen = torch.load(exp / 'checkpoint' / 'I3_E3_gen.pth')
    assert 'denoise_fn.downs.0.weight' in gen and tuple(gen['denoise_fn.downs.0.weight'].shape) == (64, 6, 3, 3)
    opt_state = torch.load(exp / 'checkpoint' / 'I3_E3_opt.pth')
    assert opt_state['iter'] == 3 and opt_state['epoch'] == 3 and 'optimizer' in opt_state
    for step in (2, 4):
        names = sorted(p.name for p in (exp / 'results' / str(step)).iterdir())
        assert names == ['%d_1_%s.png' % (step, k) for k in ('hr', 'inf', 'lr', 'sr')], names


def test_sr_py_val_runs_to_its_end_on_standins(tmp_path):
    cfg = _config(tmp_path, 'sr_sr3_16_128.json')
    rc, out, work = _run(tmp_path, 'sr.py', ['-p', 'val', '-c', cfg, '-debug'], '--force-cpu', '--standins')
    assert rc == 0, out[-3000:]
    assert '# Validation # PSNR:' in out and '# Validation # SSIM:' in out
    assert "CALL: ('reverse_loop', (1, 3, 128, 128), 10, True)" in out


def test_sample_py_unconditional_sr3_runs_to_its_end_on_standins(tmp_path):
    """config/sample_sr3_128.json: which_model_G sr3, conditional false, in_channel 3 (sample.py:104, 140)."""
    cfg = _config(tmp_path, 'sample_sr3_128.json', n_iter=2)
    rc, out, work = _run(tmp_path, 'sample.py', ['-p', 'train', '-c', cfg, '-debug'], '--force-cpu', '--standins')
    assert rc == 0, out[-3000:]
    assert out.count("CALL: ('train_step', (1, 3, 128, 128))") == 2
    # -debug: val data_len 3 samples at val_freq 2, sample(continous=False) -> ret_img[-1] (sr3 diffusion.py:180-187)
    assert out.count("CALL: ('reverse_loop', (1, 3, 128, 128), 10, False)") == 3
    names = sorted(p.name for p in (_exp_dir(work) / 'results' / '2').iterdir())
    assert names == ['2_0_sr.png', '2_1_sr.png', '2_2_sr.png'], names
