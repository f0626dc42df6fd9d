"""Runs one of the reference's own scripts, UNCHANGED, against the MI355X implementation:

    python -m ss_asr_amd.run_reference /path/to/ss_asr/src/train.py ASRTrainer exp1 conf/default.yaml runs/ result/

    # 8 GPUs of one node (plain data parallel; gradients averaged over RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        -m ss_asr_amd.run_reference /path/to/ss_asr/src/train.py ASRTrainer exp1 conf/default.yaml runs/ result/

The reference's scripts import their siblings by bare module name (`import trainer`,
src/train.py:9; `from asr import ASR`, src/trainer.py:20).  `python src/train.py` puts src/ at
sys.path[0], AHEAD of PYTHONPATH, so an environment variable cannot redirect those imports.
This launcher binds those names to this package's modules first (ss_asr_amd/flat.py installs them in
sys.modules, which an import consults before any path) and executes the script with runpy as
`__main__`; nothing in the reference tree is edited and none of its other modules is imported.
"""
import os
import runpy
import sys


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] in ('-h', '--help'):
        print(__doc__)
        return 2
    script = os.path.abspath(argv[0])
    if not os.path.isfile(script):
        print('run_reference: no such script: %s' % script, file=sys.stderr)
        return 2
    # the flat names first; the script's own directory is NOT searched (runpy.run_path does not add it)
    from . import flat  # noqa: F401
    sys.path[:] = [p for p in sys.path if os.path.abspath(p or '.') != os.path.dirname(script)]
    sys.argv = [script] + argv[1:]
    runpy.run_path(script, run_name='__main__')
    return 0


if __name__ == '__main__':
    sys.exit(main())
