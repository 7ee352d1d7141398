"""bench.py launcher contract (CPU): `--gpus N` without a launcher re-executes the command under torch.distributed.run with N ranks
on 127.0.0.1; with a launcher present it never re-spawns."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_flag_spawns_that_many_ranks(monkeypatch):
    sys.path.insert(0, ROOT)
    bench = importlib.import_module('bench')
    seen = {}

    def fake_call(cmd, env=None):
        seen['cmd'], seen['env'] = cmd, env
        return 0
    import subprocess
    monkeypatch.setattr(subprocess, 'call', fake_call)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '7', '--warmup', '2'])
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        monkeypatch.delenv(k, raising=False)
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 0
    cmd = seen['cmd']
    assert cmd[1:3] == ['-m', 'torch.distributed.run']
    assert cmd[cmd.index('--nproc-per-node') + 1] == '4' and cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert cmd[-6:] == ['--gpus', '4', '--steps', '7', '--warmup', '2'] and cmd[-7].endswith('bench.py')
    assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
