"""tracker.json bookkeeping, same file format as src/TrackerHandler.py:
{module_id: {"best": float, "step": int}}.  The reference rewrites the file on
every step; here writes can be batched with ``flush_every`` (default 1 keeps
the reference behaviour)."""
import json
import os


class TrackerHandler:
    def __init__(self, path, module_id, flush_every=1):
        self.path = path
        self.module_id = module_id
        self.flush_every = max(1, int(flush_every))
        if not os.path.exists(self.path):
            with open(self.path, 'w') as f:
                f.write('{}')
        with open(self.path, 'r') as f:
            self.data = json.load(f)
        if self.module_id not in self.data:
            self.data[self.module_id] = {'best': 10000, 'step': 0}
        self.step = self.data[self.module_id]['step']

    def do_step(self):
        self.data[self.module_id]['step'] += 1
        self.step += 1
        if self.step % self.flush_every == 0:
            self._save()

    def get_best(self):
        return self.data[self.module_id]['best']

    def set_best(self, val):
        self.data[self.module_id]['best'] = val
        self._save()

    def _save(self):
        with open(self.path, 'w') as f:
            json.dump(self.data, f)
