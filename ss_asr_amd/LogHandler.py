"""Scalar / text / image logging with the call surface of src/LogHandler.py.
tensorboardX is used when importable; otherwise records go to
<logdir>/events.jsonl (images are summarised by shape)."""
import json
import os


class LogHandler:
    def __init__(self, logdir, module_id):
        self.module_id = module_id
        self.logdir = logdir
        os.makedirs(logdir, exist_ok=True)
        try:
            from tensorboardX import SummaryWriter
            self.log = SummaryWriter(logdir)
            self._jsonl = None
        except ImportError:
            self.log = None
            self._jsonl = open(os.path.join(logdir, 'events.jsonl'), 'a')

    def _key(self, key):
        return '{}_{}'.format(self.module_id, key)

    def _write(self, kind, key, value, step):
        self._jsonl.write(json.dumps({'kind': kind, 'key': self._key(key), 'value': value,
                                      'step': int(step)}) + '\n')
        self._jsonl.flush()

    def scalar(self, key, val, step):
        if self.log is not None:
            self.log.add_scalar(self._key(key), val, step)
        else:
            self._write('scalar', key, float(val), step)

    def text(self, key, val, step):
        if self.log is not None:
            self.log.add_text(self._key(key), val, step)
        else:
            self._write('text', key, str(val), step)

    def image(self, key, val, step):
        if self.log is not None:
            self.log.add_image(self._key(key), val, step)
        else:
            self._write('image', key, list(val.shape), step)

    def figure(self, key, val, step):
        """src/LogHandler.py:26-27 (SAETrainer.valid's spectrogram comparisons); without tensorboardX the
        record is the pair of array shapes handed over."""
        if self.log is not None:
            self.log.add_figure(self._key(key), val, step)
        else:
            self._write('figure', key, [list(getattr(v, 'shape', ())) for v in (val if isinstance(val, (list, tuple)) else [val])], step)

    def embedding(self, key, val, meta, step):
        """src/LogHandler.py:29-30 (ADVTrainer.valid's real / fake frames)."""
        if self.log is not None:
            self.log.add_embedding(val, metadata=meta, tag=self._key(key), global_step=step)
        else:
            self._write('embedding', key, {'shape': list(val.shape), 'meta': {m: meta.count(m) for m in set(meta)}}, step)
