"""The C ABI without a GPU: the library builds, loads, exports every symbol include/dcunet.h declares, sizes
its workspaces, and the product path refuses to run (loudly) when there is no GPU -- no CPU fallback exists."""
import ctypes
import os
import re

import pytest


def test_library_exports_every_declared_symbol(dclib):
    from deep_calcium_amd import _lib
    protos = _lib.parse_header()
    assert len(protos) >= 36
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in protos:
        assert hasattr(raw, name), 'libdcunet.so does not export %s' % name
    # and nothing dc_* is exported that the header does not declare
    import subprocess
    out = subprocess.run(['nm', '-D', '--defined-only', _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r' T (dc_\w+)', out))
    assert exported == set(protos), exported ^ set(protos)


def test_library_never_allocates_or_synchronises():
    """dcunet.h: 'the library never allocates, frees or retains memory' and every launch is asynchronous -- no hipMalloc /
    hipFree / hipStreamSynchronize / hipDeviceSynchronize anywhere under csrc/ (round 4 had a library-owned split-K scratch)."""
    root = os.path.join(os.path.dirname(__file__), '..', 'deep_calcium_amd', 'csrc')
    for f in sorted(os.listdir(root)):
        src = open(os.path.join(root, f)).read()
        src = re.sub(r'//.*', '', src)
        src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
        for word in ('hipMalloc', 'hipFree', 'hipStreamSynchronize', 'hipDeviceSynchronize', 'hipMallocAsync', 'hipHostMalloc'):
            assert word not in src, '%s calls %s' % (f, word)


def test_abi_version_matches_the_header(dclib):
    from deep_calcium_amd import _lib
    assert dclib.dc_version() == _lib.header_abi_version() >= 102
    assert dclib.dc_conv3x3_splitk_ws_floats(20, 16, 16, 256, 256, 0) == 3 * 20 * 16 * 16 * 256      # 160 workgroups x 3 slabs
    assert dclib.dc_conv3x3_splitk_ws_floats(16, 512, 512, 32, 32, 0) == 0
    assert dclib.dc_convT2x2_dgrad_splitk_ws_floats(20, 8, 8, 512, 256) >= 0


def test_tape_trampolines_are_current_and_cover_the_launch_entry_points():
    """csrc/tape_tramp.inc is generated from include/dcunet.h (deep_calcium_amd/_gen_tape.py, run by the build): the committed copy
    must be what the generator renders from the committed header, and every launch entry point must be replayable -- except the
    ones that take host data read at call time or return through an out-parameter."""
    from deep_calcium_amd import _gen_tape, _lib
    assert open(_gen_tape.OUT).read() == _gen_tape.render()
    tapeable = set(n for n, _ in _gen_tape.prototypes())
    protos = _lib.parse_header()
    launches = set(n for n, (rt, _, _) in protos.items() if rt is ctypes.c_int and not n.endswith(('_tiles', '_blocks', '_floats', '_rows'))
                   and n != 'dc_version')
    assert launches - tapeable == {'dc_crop_augment', 'dc_event_create', 'dc_event_create_sync', 'dc_event_create_fenced', 'dc_event_elapsed_ms',
                                   'dc_event_destroy', 'dc_host_nf_pairs', 'dc_host_label8', 'dc_tape_create', 'dc_tape_destroy', 'dc_tape_append',
                                   'dc_tape_patch', 'dc_tape_replay', 'dc_tape_len', 'dc_comm_unique_id', 'dc_comm_init_rank',
                                   'dc_comm_destroy', 'dc_bracket_next_launch'}, launches - tapeable
    assert {'dc_conv3x3_fwd_f16x3', 'dc_event_record', 'dc_stream_wait_event', 'dc_adam_step_flat', 'dc_conv3x3_bwd_joint_f16x3',
            'dc_comm_all_reduce_sum', 'dc_comm_all_reduce_sum_f64', 'dc_comm_group_start', 'dc_comm_group_end'} <= tapeable      # the gradient exchange replays with the step


def test_header_cites_reference_call_sites():
    src = open(os.path.join(os.path.dirname(__file__), '..', 'include', 'dcunet.h')).read()
    assert 'unet_2d_summary.py:123-224' in src and 'unet_2d_summary.py:164-165' in src and ':156-157' in src


def test_host_side_helpers_run_without_gpu(dclib):
    L = dclib
    assert L.dc_version() >= 102
    assert L.dc_conv3x3_tiles(16, 512, 512, 32) == 16 * 32 * 16          # 512px x 32col tiles
    assert L.dc_conv3x3_tiles(16, 256, 256, 64) == 16 * 32 * 8           # 256px x 64col tiles
    assert L.dc_convT2x2_tiles(2, 32, 32, 256) == 2 * 4
    assert L.dc_conv3x3_wgrad_ws_floats(16, 512, 512, 32, 32) > 9 * 32 * 32
    assert L.dc_conv3x3_wgrad_ws_floats(2, 32, 32, 1, 32) > 0
    assert L.dc_bn_bwd_blocks(16 * 512 * 512, 32) == 1280 and L.dc_head_blocks(100) == 1


def test_argument_validation_returns_codes(dclib):
    from deep_calcium_amd._lib import DcunetError
    with pytest.raises(DcunetError, match='null pointer'):
        dclib.dc_conv3x3_fwd(None, None, None, None, 32, None, None, None, 0, 1, 8, 8, 8, 8, None)
    with pytest.raises(DcunetError, match='power of two'):
        dclib.dc_bn_relu_drop_fwd(1, 1, 1, 1, 1, None, 1.0, 0, 16, 24, 10, 24, 0.0, None, None)
    with pytest.raises(DcunetError, match='even'):
        dclib.dc_maxpool2x2_fwd(16, 8, 16, None, 1, 7, 8, 8, None)


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from deep_calcium_amd import unet_hip
    from deep_calcium_amd._lib import DcunetError
    with pytest.raises(DcunetError, match='no CPU fallback'):
        unet_hip((32, 32))


def test_product_never_imports_oracle():
    root = os.path.join(os.path.dirname(__file__), '..', 'deep_calcium_amd')
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', src, flags=re.M), f


def test_role_split_kernels_stay_below_the_sgpr_spill_cliff(tmp_path):
    """hipcc 7.0 / gfx950: with three more epilogue variants compiled into igemm_pp_kernel the compiler spilled 186-196
    SGPRs to VGPR lanes (60-64 today) and the kernel computed wrong, run-to-run different BatchNorm partials at batch 16
    (DESIGN 5c).  The variants are separate instantiations now; this keeps an edit from walking back over that cliff
    unnoticed on the CPU box (the full-size GPU tests are the real check)."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('hipcc not available')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, 'deep_calcium_amd', 'csrc', 'igemm_pp.hip')
    out = str(tmp_path / 'pp.s')
    r = subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-x', 'hip', '-S', '--cuda-device-only', src, '-o', out],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    text = open(out).read()
    names = re.findall(r'\.name:\s+(_Z15igemm_pp_kernel\w+)\n', text)
    spills = [int(x) for x in re.findall(r'\.sgpr_spill_count:\s+(\d+)', text)]
    scratch = [int(x) for x in re.findall(r'; ScratchSize: (\d+)', text)]
    assert len(spills) >= 4 and len(set(names)) >= 4, (names, spills)
    assert max(spills) <= 72, spills            # verified good at 53-64; broken at 186-196
    assert all(s == 0 for s in scratch), scratch


def test_library_loads_without_rccl_and_says_so_when_asked(dclib):
    """librccl is resolved with dlopen at the first dc_comm_* call (csrc/comm.cpp): the library itself carries no link-time dependency on it
    (a box with nothing that links RCCL must still run every other entry point), and the collective entry points exist in the ABI."""
    import subprocess
    from deep_calcium_amd import _build, _lib
    out = subprocess.run(['readelf', '-d', _build.LIB], capture_output=True, text=True).stdout
    needed = [l.split('[')[-1].rstrip(']') for l in out.splitlines() if 'NEEDED' in l]
    assert needed and not [n for n in needed if 'rccl' in n.lower() or 'nccl' in n.lower()], needed
    protos = _lib.parse_header()
    for name in ('dc_comm_unique_id', 'dc_comm_init_rank', 'dc_comm_all_reduce_sum', 'dc_comm_all_reduce_sum_f64', 'dc_comm_group_start',
                 'dc_comm_group_end', 'dc_comm_destroy', 'dc_event_create_fenced', 'dc_bracket_next_launch'):
        assert name in protos and hasattr(dclib.cdll, name), name
    # bad arguments are rejected before RCCL is even looked for
    assert dclib.cdll.dc_comm_all_reduce_sum(None, None, 0, None) == -1
    assert b'dc_comm_all_reduce_sum' in dclib.cdll.dc_last_error()


def test_import_raises_the_hardware_queue_count_only_when_unset():
    """deep_calcium_amd/__init__.py sets GPU_MAX_HW_QUEUES=8 as a DEFAULT (with HIP's 4 queues a data-parallel rank's
    weight-gradient stream can land on the main stream's hardware queue and the two-stream backward serialises); an explicit value wins."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "import os, deep_calcium_amd; print(os.environ['GPU_MAX_HW_QUEUES'])"
    for given, want in ((None, '8'), ('4', '4'), ('16', '16')):
        env = dict(os.environ, PYTHONPATH=root)
        env.pop('GPU_MAX_HW_QUEUES', None)
        if given is not None:
            env['GPU_MAX_HW_QUEUES'] = given
        out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0 and out.stdout.strip() == want, (given, out.stdout, out.stderr[-300:])
