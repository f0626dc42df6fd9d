"""Solver / ASRTrainer with the surface of the reference's src/trainer.py
(Solver :33-195, ASRTrainer :374-545, TAETrainer :594-758, SAETrainer :760-907, ADVTrainer :909-1124,
asr_seed_train :1126-1177) so that ``src/train.py`` drives it unchanged:
``getattr(trainer, 'ASRTrainer')(config, paras)`` then ``load_data()``, ``set_model()``, ``exec()``.

Differences, all inside the same call surface:
  * the model computes through libssasr_hip.so (MI355X only);
  * with Adadelta (conf/default.yaml) one iteration of exec() is ONE call of
    engine.ASRTrainStep -- the object bench.py times: zero_grad folded into the
    update kernel, forward, masked CE, backward with the weight gradients on a
    second stream, two-bucket gradient all-reduce, fused clip + NaN guard +
    Adadelta (optim.py), no device-to-host copy -- fed by the device-resident
    loader (gpu_loader.py).  Any other optimizer type takes the reference's
    sequence (zero_grad, forward, backward, Solver.step with clip_grad_norm_);
  * launched under torchrun (WORLD_SIZE > 1) rank r takes batches r, r + world,
    ... of every full round of `world` batches (a tail that does not fill a round
    is dropped, so every rank runs the same number of steps) and gradients are
    averaged over RCCL (dist.py); rank 0 alone logs and checkpoints;
  * valid() no longer dies on the undefined names of src/trainer.py:531.
"""
import math
import warnings
import os

import numpy as np
import torch
import torch.nn as nn

from . import dist as sdist
from . import ops
from .ASRDataset import load_asr_dataset, prepare_x, prepare_y
from .LogHandler import LogHandler
from .TrackerHandler import TrackerHandler
from .asr import ASR
from .optim import FusedAdadelta
from .postprocess import calc_acc, calc_err, draw_att


class Solver:
    def __init__(self, config, paras, module_id):
        self.config = config
        self.paras = paras
        self.module_id = module_id
        self.rank, self.world, local = sdist.init_from_env()

        if torch.cuda.is_available():
            torch.cuda.set_device(local)
            self.device = torch.device('cuda', local)
            self.verbose("A cuda device is available and will be used.")
            self.paras.gpu = True
        else:
            # Kept for the plumbing tests; the model's forward refuses CPU tensors.
            self.device = torch.device('cpu')
            self.verbose("No cuda device available.")
            self.paras.gpu = False

        os.makedirs(paras.ckpdir, exist_ok=True)
        self.ckpdir = os.path.join(self.paras.ckpdir, self.paras.name)
        os.makedirs(self.ckpdir, exist_ok=True)
        suffix = '' if self.rank == 0 else '.rank%d' % self.rank
        self.tr = TrackerHandler(os.path.join(self.ckpdir, 'tracker%s.json' % suffix),
                                 self.module_id)
        self.lg = LogHandler(os.path.join(self.paras.logdir, self.paras.name,
                                          self.module_id + suffix), self.module_id)
        self.ckppath = os.path.join(self.ckpdir, self.module_id + '.cpt')
        self.best_ckppath = os.path.join(self.ckpdir, self.module_id + '_best.cpt')

        self.valid_step = self.set_if_exists('valid_step', 500)
        self.logging_step = self.set_if_exists('logging_step', 250)
        self.save_step = self.set_if_exists('save_step', 1000)
        self.n_epochs = self.set_if_exists('n_epochs', 5)
        self.train_batch_size = self.set_if_exists('train_batch_size', 32)
        self.valid_batch_size = self.set_if_exists('valid_batch_size', 32)
        self.test_batch_size = self.set_if_exists('test_batch_size', 1)
        self.verbose_summary()

    def verbose_summary(self):
        self.verbose("-------SUMMARY-------")
        self.verbose("Current step : {}".format(self.tr.step))
        self.verbose("Best metric value : {}".format(self.tr.get_best()))
        self.verbose("Number of epochs: {}".format(self.n_epochs))
        self.verbose("Steps: [Logging {}], [Saving {}], [Validation {}]".format(
            self.logging_step, self.save_step, self.valid_step))
        self.verbose("Batch sizes: [Train {}], [Validation{}], [Testing {}]".format(
            self.train_batch_size, self.valid_batch_size, self.test_batch_size))
        self.verbose("---------------------")

    def set_if_exists(self, key, default):
        return self.config[self.module_id].get(key, default)

    def verbose(self, msg, progress=False):
        end = '\r' if progress else '\n'
        if progress:
            msg += '                              '
        else:
            msg = '[INFO ({} / {})] '.format(self.module_id, self.paras.name) + str(msg)
        if self.paras.verbose and getattr(self, 'rank', 0) == 0:
            print(msg, end=end)

    def step(self, params, optim, grad_clip=5):
        """src/trainer.py:131-148: clip the global gradient norm, skip the update
        (with a message) when the norm is NaN, otherwise step the optimizer.
        With FusedAdadelta everything happens on the device; the NaN message of
        a step is printed when its flag has reached the host (at the latest on
        the next call)."""
        if isinstance(optim, FusedAdadelta):
            done = optim.poll()
            if done is not None and done[1]:
                self.verbose('Error : grad norm is NaN @ step {}'.format(self.tr.step - 1))
            scale = sdist.allreduce_grad(optim._grad)
            optim.clip_and_step(max_norm=float(grad_clip), grad_scale=scale)
            return
        # Any other torch.optim type a config names (conf/default.yaml selects none): the reference's own sequence on
        # torch's optimizer -- NOT the MI355X-native step (no fused clip + update kernel, no overlapped all-reduce).
        params = list(params)
        if not getattr(self, '_warned_torch_optim', False):
            self._warned_torch_optim = True
            warnings.warn('%s has no fused MI355X form here: Solver.step runs torch.optim and clip_grad_norm_ '
                          '(Adadelta and Adam are the native ones)' % type(optim).__name__, RuntimeWarning)
        if sdist.is_active():
            # every parameter the optimizer updates is averaged, not only the list the norm is clipped over
            # (TAETrainer / SAETrainer hand `params` of ONE module to an optimizer over two: src/trainer.py:633-641, :676)
            seen = set()
            for grp in optim.param_groups:
                for p in grp['params']:
                    if p.grad is not None and id(p) not in seen:
                        seen.add(id(p))
                        torch.distributed.all_reduce(p.grad)
                        p.grad.div_(sdist.world_size())
        grad_norm = nn.utils.clip_grad_norm_(params, grad_clip)
        if math.isnan(grad_norm):
            self.verbose('Error : grad norm is NaN @ step {}'.format(self.tr.step))
        else:
            optim.step()

    def setup_module(self, module, ckp_path, *para, **model_para):
        model = module(*para, **model_para)
        if os.path.isfile(ckp_path):
            self.verbose('Loading a pretrained model from {}'.format(ckp_path))
            model.load_state_dict(torch.load(ckp_path, map_location='cpu'))
        else:
            self.verbose('No model found at {}. A new model will be created'.format(ckp_path))
        return model.to(self.device)

    def genpath(self, p, module_id):
        if p is None:
            path = os.path.join(self.ckpdir, '{}.cpt'.format(module_id))
            return (path, path)
        if isinstance(p, str):
            return (p, p)
        assert len(p) == 2
        return p

    def close(self):
        return None


class ASRTrainer(Solver):
    def __init__(self, config, paras):
        super().__init__(config, paras, 'asr')

    def load_data(self):
        """Index files must be sorted so that every batch is in decreasing
        frame-length order (conf/README.md:16)."""
        n_jobs = self.set_if_exists('loader_jobs', 8)
        (self.mapper, _, self.train_set) = load_asr_dataset(
            self.config['asr']['train_index'], batch_size=self.train_batch_size,
            n_jobs=n_jobs, use_gpu=self.paras.gpu)
        (_, _, self.valid_set) = load_asr_dataset(
            self.config['asr']['valid_index'], batch_size=self.valid_batch_size,
            n_jobs=n_jobs, use_gpu=self.paras.gpu)
        self.wer_step = self.config['asr']['wer_step']
        # Training batches come from a device-resident copy of the corpus when it fits
        # comfortably; ``asr.gpu_resident_loader: false`` keeps the DataLoader.
        self.gpu_loader = None
        if self.device.type == 'cuda' and self.config['asr'].get('gpu_resident_loader', True):
            from .gpu_loader import GpuResidentLoader
            free, _ = torch.cuda.mem_get_info(self.device)
            est = 4 * int(self.config['asr']['mdl']['feature_dim']) * sum(
                r['unpadded_num_frames'] for r in self.train_set.dataset._rows)
            if est < free // 4:
                self.gpu_loader = GpuResidentLoader(self.config['asr']['train_index'], self.train_batch_size,
                                                    self.device, rank=self.rank, world=self.world)
                self.verbose('Training corpus resident on the GPU: {:.1f} MB'.format(
                    self.gpu_loader.bytes_resident() / 1e6))

    def _train_batches(self):
        """(batch index, x, x_lens, y, y_lens) for this rank's batches of one epoch.  Every rank
        yields the same number of batches: a tail that does not fill a round of `world` batches
        is dropped (a rank alone in an all-reduce would wait for ever)."""
        if self.gpu_loader is not None:
            yield from self.gpu_loader
            return
        from .gpu_loader import rank_batches
        mine = set(rank_batches(len(self.train_set), self.rank, self.world))
        for b_ind, (x, y) in enumerate(self.train_set):
            if b_ind not in mine:
                continue
            (x, x_lens) = prepare_x(x, device=self.device)
            (y, y_lens) = prepare_y(y, device=self.device)
            yield b_ind, x, x_lens, y, y_lens

    def set_model(self, asrpath=None):
        """asrpath (this build; the Seed loop passes it): (checkpoint to load, checkpoint to save) instead of
        <ckpdir>/asr.cpt for both, as the other trainers of the reference take it (src/trainer.py:623)."""
        if asrpath is not None:
            self.ckppath_in, self.ckppath = self.genpath(asrpath, 'asr')
        # `ctc_weight` under asr.mdl is a key of this build (BASELINE.json configs[3], ss_asr_amd/ctc.py);
        # the reference's configs do not carry it and get the reference's model and loss
        joint = 'ctc_weight' in self.config['asr']['mdl']
        if joint:
            from .ctc import JointCTCASR, JointCTCTrainStep
        self.asr_model = self.setup_module(JointCTCASR if joint else ASR, getattr(self, 'ckppath_in', self.ckppath),
                                           self.mapper.get_dim(),
                                           **self.config['asr']['mdl'])
        opt = self.config['asr']['opt']
        self.train_step = None
        if opt['type'] == 'Adadelta' and self.device.type == 'cuda':
            # the fused step (engine.ASRTrainStep): what bench.py times is what trains
            from .engine import ASRTrainStep
            step_cls = JointCTCTrainStep if joint else ASRTrainStep
            self.train_step = step_cls(self.asr_model, lr=opt['learning_rate'], eps=1e-8, grad_clip=5.0)
            self.flat, self.optim = self.train_step.flat, self.train_step.optim
        elif joint:
            raise RuntimeError('the joint CTC+attention loss runs on the fused step only (Adadelta on the GPU)')
        else:
            self.optim = getattr(torch.optim, opt['type'])(
                self.asr_model.parameters(), lr=opt['learning_rate'], eps=1e-8)

    def _loss(self, prediction, y, ans_len):
        """src/trainer.py:426-434."""
        if prediction.is_cuda:
            return ops.masked_ce_loss(prediction, y, ans_len)
        raise RuntimeError('ss_asr_amd computes on the GPU only')

    def exec(self):
        self.verbose('Training set total {} batches'.format(len(self.train_set)))
        self._nan_reported = self.train_step.skipped_steps if self.train_step is not None else 0
        epoch = 0
        while epoch < self.n_epochs:
            self.verbose("Starting epoch {} out of {}".format(epoch + 1, self.n_epochs))
            for b_ind, x, x_lens, y, y_lens in self._train_batches():
                self.verbose('Batch: {}/{}, global step: {}'.format(
                    b_ind, len(self.train_set), self.tr.step), progress=True)
                state_len = x_lens
                ans_len = max(y_lens) - 1

                if self.train_step is not None:
                    # src/trainer.py:419-438 as one fused step; a persistent launch that timed out
                    # in the previous step raises here (its status words arrive with the optimizer's)
                    loss = self.train_step(x, y, state_len, ans_len)
                    prediction = self.train_step.last_logits
                    if self.train_step.skipped_steps > self._nan_reported:
                        self._nan_reported = self.train_step.skipped_steps
                        self.verbose('Error : grad norm is NaN @ step {}'.format(self.tr.step - 1))
                else:
                    self.optim.zero_grad()
                    _, prediction, _ = self.asr_model(x, ans_len, teacher=y, state_len=state_len)
                    loss = self._loss(prediction, y, ans_len)
                    loss.backward()
                    self.step(self.asr_model.parameters(), self.optim)
                    if self.tr.step % self.logging_step == 0:
                        ops.check_persistent_status()      # the host synchronises here anyway
                label = y[:, 1:ans_len + 1]
                if self.rank == 0:
                    if self.tr.step % self.logging_step == 0:
                        self.lg.scalar('train_loss', loss.item(), self.tr.step)
                        self.lg.scalar('train_acc', calc_acc(prediction, label), self.tr.step)
                    if self.tr.step % self.wer_step == 0:
                        self.lg.scalar('train_error',
                                       calc_err(prediction, label, mapper=self.mapper),
                                       self.tr.step)
                    if self.tr.step % self.save_step == 0:
                        if self.train_step is not None:
                            self.train_step.finish()   # this step's verdict first: never checkpoint after a time-out
                        self.verbose("Model saved at step {}".format(self.tr.step))
                        sdist.save_atomic(self.asr_model.state_dict(), self.ckppath)
                if self.tr.step % self.valid_step == 0:
                    self.valid()
                self.tr.do_step()
            epoch += 1
        if self.train_step is not None:
            self.train_step.finish()           # the last step's verdict (raises on a timeout)

    def valid(self):
        """Greedy decoding for ans_len + 30 steps without a teacher, loss on the
        first ans_len outputs (src/trainer.py:460-537)."""
        self.asr_model.eval()
        total_loss, total_acc, total_err, num_batches = 0.0, 0.0, 0.0, 0
        prediction = label = att_map = None
        # Under torchrun the validation batches are dealt to the ranks round-robin (every rank holds the same
        # weights: the sums below are all-reduced, the averages are the single-process ones); rank 0 also takes
        # the LAST batch, whose hypotheses it logs as the reference does (src/trainer.py:505-519).
        n_valid = len(self.valid_set)
        mine = lambda b: self.world == 1 or b % self.world == self.rank or (self.rank == 0 and b == n_valid - 1)
        counted = lambda b: self.world == 1 or b % self.world == self.rank
        with torch.no_grad():
            for b_idx, (x, y) in enumerate(self.valid_set):
                if not mine(b_idx):
                    continue
                self.verbose('Validation step - ( {} / {} )'.format(b_idx, len(self.valid_set)),
                             progress=True)
                (x, x_lens) = prepare_x(x, device=self.device)
                (y, y_lens) = prepare_y(y, device=self.device)
                ans_len = max(y_lens) - 1
                _, prediction, att_map = self.asr_model(x, ans_len + 30, state_len=x_lens)
                label = y[:, 1:ans_len + 1].contiguous()
                if not counted(b_idx):
                    continue                       # (rank 0's copy of the last batch: for its log only)
                loss = self._loss(prediction, y, ans_len)
                total_loss += float(loss)
                total_acc += calc_acc(prediction, label)
                total_err += calc_err(prediction, label, mapper=self.mapper)
                num_batches += 1
        # the persistent launches of the greedy passes report into status words of their own (they run
        # outside a train step's shared row): a hand-off that timed out must not become a "best" model
        # or a best_hyp.txt (float(loss) above has synchronised already)
        ops.check_persistent_status()
        if sdist.is_active() and self.world > 1:
            sums = torch.tensor([total_loss, total_acc, total_err, float(num_batches)], dtype=torch.float64,
                                device=self.device)
            torch.distributed.all_reduce(sums)
            total_loss, total_acc, total_err, num_batches = (float(sums[0]), float(sums[1]), float(sums[2]),
                                                             int(round(float(sums[3]))))
        if num_batches == 0:
            self.asr_model.train()
            return
        avg_loss = total_loss / num_batches
        avg_err = total_err / num_batches
        avg_acc = total_acc / num_batches
        if self.rank == 0:
            self.lg.scalar('eval_loss', avg_loss, self.tr.step)
            self.lg.scalar('eval_error', avg_err, self.tr.step)
            self.lg.scalar('eval_acc', avg_acc, self.tr.step)
            hyp_idx = np.argmax(prediction.cpu().numpy(), axis=-1)
            val_hyp = [self.mapper.translate(p) for p in hyp_idx]
            val_txt = [self.mapper.translate(l) for l in label.cpu()]
            for idx, attmap in enumerate(draw_att(att_map, hyp_idx)):
                self.lg.image('eval_att_' + str(idx), attmap, self.tr.step)
                self.lg.text('eval_hyp_' + str(idx), "{} |predict vs. real| {}".format(
                    val_hyp[idx], val_txt[idx]), self.tr.step)
            if avg_loss < self.tr.get_best():
                self.tr.set_best(avg_loss)
                self.verbose('Best validation loss for ASR : {:.4f} @ global step {}'.format(
                    self.tr.get_best(), self.tr.step))
                self.verbose('Saving best model.')
                sdist.save_atomic(self.asr_model.state_dict(), self.best_ckppath)
                with open(os.path.join(self.ckpdir, 'best_hyp.txt'), 'w') as f:
                    for hyp, txt in zip(val_hyp, val_txt):
                        f.write(hyp + ',' + txt + '\n')
            else:
                self.verbose("Validation metric worse : ({:.4f} vs. {:.4f})".format(
                    avg_loss, self.tr.get_best()))
        self.asr_model.train()

    def close(self):
        self.verbose("Finished training! The most recent model will" +
                     "be saved at step {}".format(self.tr.step))
        if getattr(self, 'train_step', None) is not None:
            self.train_step.finish()
        if self.rank == 0:
            sdist.save_atomic(self.asr_model.state_dict(), self.ckppath)
        sdist.barrier()        # the next reader of these files (the Seed loop's next leg) runs on every rank


class TAETrainer(Solver):
    """Trains the text autoencoder, and through it the ASR model's attention / speller / embedding /
    char_trans (src/trainer.py:594-758; config 5's first leg).  With Adam (conf/default.yaml:43-45) on the
    GPU one iteration of exec() is ONE engine.TAETrainStep call; any other optimizer type takes the
    reference's sequence (zero_grad, forward, backward, Solver.step over the text autoencoder's parameters)."""

    def __init__(self, config, paras):
        super().__init__(config, paras, 'tae')

    def load_data(self):
        """Text only, with noise: the loaders yield (clean_y, noised_y) (src/trainer.py:601-614)."""
        (self.mapper, self.dataset, self.train_set) = load_asr_dataset(
            self.config['tae']['train_index'], batch_size=self.train_batch_size, use_gpu=self.paras.gpu,
            text_only=True, drop_rate=self.config['tae']['drop_rate'],
            n_jobs=self.set_if_exists('loader_jobs', 8))
        (_, _, self.valid_set) = load_asr_dataset(
            self.config['tae']['valid_index'], batch_size=self.valid_batch_size, use_gpu=self.paras.gpu,
            text_only=True, drop_rate=self.config['tae']['drop_rate'],
            n_jobs=self.set_if_exists('loader_jobs', 8))

    def set_model(self, asrpath=None, asr_model=None):
        """src/trainer.py:616-644.  asr_model (this build): an ASR object that is already in memory -- the
        Seed loop's legs then train the SAME parameters in turn without going through a checkpoint."""
        from .text_autoencoder import TextAutoEncoder
        (self.asrpath_in, self.asrpath_out) = self.genpath(asrpath, 'asr')
        self.asr_model = asr_model if asr_model is not None else self.setup_module(
            ASR, self.asrpath_in, self.mapper.get_dim(), **self.config['asr']['mdl'])
        self.text_autoenc = self.setup_module(TextAutoEncoder, self.ckppath, self.mapper.get_dim(),
                                              **self.config['tae']['mdl'])
        opt = self.config['tae']['opt']
        self.train_step = None
        if opt['type'] == 'Adam' and self.device.type == 'cuda':
            from .engine import TAETrainStep
            self.train_step = TAETrainStep(self.asr_model, self.text_autoenc, lr=opt['learning_rate'], eps=1e-8,
                                           grad_clip=5.0)
            self.optim = self.train_step.optim
        else:
            # the optimizer steps the text autoencoder, the ASR character embedding, attention module,
            # speller and char_trans layer (src/trainer.py:625-641)
            self.optim = getattr(torch.optim, opt['type'])(
                list(self.text_autoenc.parameters()) + list(self.asr_model.embed.parameters()) +
                list(self.asr_model.attention.parameters()) + list(self.asr_model.decoder.parameters()) +
                list(self.asr_model.char_trans.parameters()), lr=opt['learning_rate'], eps=1e-8)

    def _loss(self, enc_out, y):
        """src/trainer.py:662-672."""
        from .text_autoencoder import tae_loss
        if not enc_out.is_cuda:
            raise RuntimeError('ss_asr_amd computes on the GPU only')
        return tae_loss(enc_out, y)

    def _rank_batches(self):
        from .gpu_loader import rank_batches
        mine = set(rank_batches(len(self.train_set), self.rank, self.world))
        for b_ind, (y, y_noise) in enumerate(self.train_set):
            if b_ind in mine:
                yield b_ind, y, y_noise

    def exec(self):
        self.verbose('Training set total {} batches'.format(len(self.train_set)))
        nan_reported = self.train_step.skipped_steps if self.train_step is not None else 0
        epoch = 0
        while epoch < self.n_epochs:
            self.verbose("Starting epoch {} out of {}".format(epoch + 1, self.n_epochs))
            for b_ind, y, y_noise in self._rank_batches():
                self.verbose('Batch: {}/{}, global step: {}'.format(
                    b_ind, len(self.train_set), self.tr.step), progress=True)
                y, y_lens = prepare_y(y, device=self.device)
                y_noise, y_noise_lens = prepare_y(y_noise, device=self.device)
                if self.train_step is not None:
                    loss = self.train_step(y, y_noise, y_lens, y_noise_lens)
                    if self.train_step.skipped_steps > nan_reported:
                        nan_reported = self.train_step.skipped_steps
                        self.verbose('Error : grad norm is NaN @ step {}'.format(self.tr.step - 1))
                else:
                    self.optim.zero_grad()
                    # decode steps == longest target
                    _, enc_out = self.text_autoenc(self.asr_model, y, y_noise, max(y_lens), noise_lens=y_noise_lens)
                    loss = self._loss(enc_out, y)
                    loss.backward()
                    self.step(self.text_autoenc.parameters(), self.optim)
                    if self.tr.step % self.logging_step == 0:
                        ops.check_persistent_status()
                if self.rank == 0 and self.tr.step % self.logging_step == 0:
                    self.lg.scalar('train_loss', loss.item(), self.tr.step)
                if self.tr.step % self.valid_step == 0:
                    self.valid()
                if self.rank == 0 and self.tr.step % self.save_step == 0:
                    if self.train_step is not None:
                        self.train_step.finish()       # never checkpoint after a time-out
                    self.verbose("Model saved at step {}".format(self.tr.step))
                    sdist.save_atomic(self.text_autoenc.state_dict(), self.ckppath)
                    sdist.save_atomic(self.asr_model.state_dict(), self.asrpath_out)
                self.tr.do_step()
            epoch += 1
        if self.train_step is not None:
            self.train_step.finish()

    def valid(self):
        """src/trainer.py:683-748."""
        self.text_autoenc.eval()
        self.asr_model.eval()
        total, n_batches = 0.0, 0
        y = enc_out = None
        with torch.no_grad():
            for b_idx, (y, y_noise) in enumerate(self.valid_set):
                self.verbose('Validation step -( {} / {} )'.format(b_idx, len(self.valid_set)), progress=True)
                y, y_lens = prepare_y(y, device=self.device)
                y_noise, y_noise_lens = prepare_y(y_noise, device=self.device)
                _, enc_out = self.text_autoenc(self.asr_model, y, y_noise, max(y_lens), noise_lens=y_noise_lens)
                total += float(self._loss(enc_out, y))
                n_batches += 1
        ops.check_persistent_status()
        self.text_autoenc.train()
        self.asr_model.train()
        if n_batches == 0:
            return
        avg_loss = total / n_batches
        if self.rank != 0:
            return
        # compare the strings of the last batch
        labels = [self.mapper.translate(l) for l in y.cpu()]
        predicts = [self.mapper.translate(p) for p in np.argmax(enc_out.cpu().numpy(), axis=-1)]
        for i in range(min(self.valid_batch_size, len(labels))):
            self.lg.text('eval_text' + str(i), '{} |vs.| {}'.format(labels[i], predicts[i]), self.tr.step)
        self.lg.scalar('eval_loss', avg_loss, self.tr.step)
        if avg_loss < self.tr.get_best():
            self.tr.set_best(avg_loss)
            self.verbose('Best validation loss : {:.4f} @ global step {}'.format(self.tr.get_best(), self.tr.step))
            sdist.save_atomic(self.text_autoenc.state_dict(), self.best_ckppath)
            self.verbose("Both the text autoencoder and ASR have been saved")
        else:
            self.verbose("Validation metric worse : ({:.4f} vs. {:.4f})".format(avg_loss, self.tr.get_best()))

    def close(self):
        self.verbose("Finished training! The most recent model will" +
                     "be saved at step {} as well as the ASR model".format(self.tr.step))
        if getattr(self, 'train_step', None) is not None:
            self.train_step.finish()
        if self.rank == 0:
            sdist.save_atomic(self.text_autoenc.state_dict(), self.ckppath)
            sdist.save_atomic(self.asr_model.state_dict(), self.asrpath_out)
        sdist.barrier()        # the next reader of these files (the Seed loop's next leg) runs on every rank


class ADVTrainer(Solver):
    """Adversarial training of the Listener against the text encoder's frames (src/trainer.py:909-1124;
    config 5's second leg).  Generator: the ASR model's Listener; data distribution: the text autoencoder's
    encoder; discriminator: discriminator.Discriminator.  One iteration of exec() is ONE engine.ADVTrainStep
    call when G_opt / D_opt are Adadelta or Adam; any other optimizer type takes the reference's sequence with
    torch optimizers (`_iteration`).

    The reference's ADVTrainer cannot run as shipped; what is fixed here, and
    nothing else: `self.loss_metric` (used at :984, never set) is nn.BCELoss -- on the ssasr_bce kernels; the
    validation index is `adv.eval_index` when the config has it (:918) and `adv.valid_index` (the key
    conf/default.yaml:74 carries) otherwise; the character-LM dataset of :922-924 (`chunk_size`,
    `lm_train_index`: absent from the yaml, and never read by exec or valid) is not loaded."""

    def __init__(self, config, paras):
        super().__init__(config, paras, 'adv')

    def load_data(self):
        n_jobs = self.set_if_exists('loader_jobs', 8)
        (self.mapper, self.dataset, self.train_set) = load_asr_dataset(
            self.config['adv']['train_index'], batch_size=self.train_batch_size, use_gpu=self.paras.gpu,
            n_jobs=n_jobs)
        eval_index = self.config['adv'].get('eval_index', self.config['adv'].get('valid_index'))
        (_, _, self.valid_set) = load_asr_dataset(eval_index, batch_size=self.valid_batch_size,
                                                  use_gpu=self.paras.gpu, n_jobs=n_jobs)

    def set_model(self, asrpath=None, taepath=None, asr_model=None, text_autoenc=None):
        """src/trainer.py:926-948.  asr_model / text_autoenc (this build): objects already in memory."""
        from .discriminator import Discriminator
        from .engine import ADVTrainStep
        from .text_autoencoder import TextAutoEncoder
        (self.asrpath_in, self.asrpath_out) = self.genpath(asrpath, 'asr')
        (taepath_in, _) = self.genpath(taepath, 'tae')
        self.asr_model = asr_model if asr_model is not None else self.setup_module(
            ASR, self.asrpath_in, self.mapper.get_dim(), **self.config['asr']['mdl'])
        self.text_autoenc = text_autoenc if text_autoenc is not None else self.setup_module(
            TextAutoEncoder, taepath_in, self.mapper.get_dim(), **self.config['tae']['mdl'])
        self.discriminator = self.setup_module(Discriminator, self.ckppath, self.asr_model.encoder.get_outdim(),
                                               **self.config['adv']['mdl'])
        self.data_distribution = self.text_autoenc.encoder
        g, d = self.config['adv']['G_opt'], self.config['adv']['D_opt']
        fused = ('Adadelta', 'Adam')
        self.train_step = None
        if g['type'] in fused and d['type'] in fused and self.device.type == 'cuda':
            self.train_step = ADVTrainStep(self.asr_model, self.text_autoenc, self.discriminator,
                                           g_opt=(g['type'], g['learning_rate']), d_opt=(d['type'], d['learning_rate']),
                                           label_smoothing=self.config['adv']['label_smoothing'], grad_clip=5.0)
            self.G_optim, self.D_optim = self.train_step.G_optim, self.train_step.D_optim
        else:
            # any other optimizer type: the reference's sequence with torch optimizers (src/trainer.py:938-948)
            self.G_optim = getattr(torch.optim, g['type'])(self.asr_model.encoder.parameters(), lr=g['learning_rate'], eps=1e-8)
            self.D_optim = getattr(torch.optim, d['type'])(self.discriminator.parameters(), lr=d['learning_rate'], eps=1e-8)

    def _frames(self, x, x_lens, y):
        """(text-encoder frames without a graph, Listener frames with theirs)."""
        if self.train_step is not None:
            return self.train_step.frames(x, x_lens, y)
        if not x.is_cuda:
            raise RuntimeError('ss_asr_amd computes on the GPU only (no CPU path)')
        with torch.no_grad():
            real = self.data_distribution(y)
        fake, _ = self.asr_model.encoder(x, x_lens)
        return real, fake

    def _iteration(self, x, x_lens, y):
        """src/trainer.py:968-1032 with torch optimizers (the fused form is engine.ADVTrainStep)."""
        from .seed_ops import bce_loss
        self.discriminator.zero_grad()
        real, fake = self._frames(x, x_lens, y)
        D_realloss = bce_loss(self.discriminator(real), 1.0 - self.config['adv']['label_smoothing'])
        D_realloss.backward()
        D_fakeloss = bce_loss(self.discriminator(fake.detach()), 0.0)
        D_fakeloss.backward()
        self.step(self.discriminator.parameters(), self.D_optim)
        self.asr_model.encoder.zero_grad()
        G_loss = bce_loss(self.discriminator(fake, frozen=True), 1.0)
        G_loss.backward()
        self.step(self.asr_model.encoder.parameters(), self.G_optim)
        if self.tr.step % self.logging_step == 0:
            ops.check_persistent_status()
        return D_realloss, D_fakeloss, G_loss

    def _rank_batches(self):
        from .gpu_loader import rank_batches
        mine = set(rank_batches(len(self.train_set), self.rank, self.world))
        for b_idx, (x, y) in enumerate(self.train_set):
            if b_idx in mine:
                yield b_idx, x, y

    def exec(self):
        self.verbose('Training set total {} batches'.format(len(self.train_set)))
        nan_reported = self.train_step.skipped_steps if self.train_step is not None else 0
        epoch = 0
        while epoch < self.n_epochs:
            self.verbose("Starting epoch {} out of {}".format(epoch + 1, self.n_epochs))
            for b_idx, x, y in self._rank_batches():
                self.verbose('Global step - {} ( {} / {} )'.format(self.tr.step, b_idx, len(self.train_set)),
                             progress=True)
                x, x_lens = prepare_x(x, device=self.device)
                y, _ = prepare_y(y, device=self.device)
                if self.train_step is not None:
                    D_realloss, D_fakeloss, G_loss = self.train_step(x, x_lens, y)
                    if self.train_step.skipped_steps > nan_reported:
                        nan_reported = self.train_step.skipped_steps
                        self.verbose('Error : grad norm is NaN @ step {}'.format(self.tr.step - 1))
                else:
                    D_realloss, D_fakeloss, G_loss = self._iteration(x, x_lens, y)
                if self.rank == 0 and self.tr.step % self.logging_step == 0:
                    self.lg.scalar('discrim_real_loss_train', D_realloss.item(), self.tr.step)
                    self.lg.scalar('discrim_fake_loss_train', D_fakeloss.item(), self.tr.step)
                    self.lg.scalar('discrim_loss_train', (D_realloss + D_fakeloss).item(), self.tr.step)
                    self.lg.scalar('gen_loss_train', G_loss.item(), self.tr.step)
                if self.tr.step % self.valid_step == 0:
                    self.valid()
                if self.rank == 0 and self.tr.step % self.save_step == 0:
                    if self.train_step is not None:
                        self.train_step.finish()       # never checkpoint after a time-out
                    self.verbose("Model saved at step {}".format(self.tr.step))
                    sdist.save_atomic(self.discriminator.state_dict(), self.ckppath)
                    sdist.save_atomic(self.asr_model.state_dict(), self.asrpath_out)
                self.tr.do_step()
            epoch += 1
        if self.train_step is not None:
            self.train_step.finish()

    def valid(self):
        """src/trainer.py:1038-1113: the discriminator's two losses (real labels unsmoothed here, :1059) averaged
        over the validation batches; the last batch's first utterance goes to the embedding log."""
        from .seed_ops import bce_loss
        self.asr_model.eval()
        self.discriminator.eval()
        real_sum = fake_sum = 0.0
        n_batches = 0
        real_data = fake_data = None
        with torch.no_grad():
            for b_idx, (x, y) in enumerate(self.valid_set):
                self.verbose('Validation step - {} ( {} / {} )'.format(self.tr.step, b_idx, len(self.valid_set)),
                             progress=True)
                x, x_lens = prepare_x(x, device=self.device)
                y, _ = prepare_y(y, device=self.device)
                real_data, fake_data = self._frames(x, x_lens, y)
                real_sum += float(bce_loss(self.discriminator(real_data), 1.0))
                fake_sum += float(bce_loss(self.discriminator(fake_data), 0.0))
                n_batches += 1
        ops.check_persistent_status()
        self.asr_model.train()
        self.discriminator.train()
        if n_batches == 0 or self.rank != 0:
            return
        avg_real, avg_fake = real_sum / n_batches, fake_sum / n_batches
        real_emb, fake_emb = real_data[0], fake_data[0]
        self.lg.embedding('validation_emb', torch.cat((real_emb, fake_emb)).cpu(),
                          ['real'] * real_emb.shape[0] + ['fake'] * fake_emb.shape[0], self.tr.step)
        avg_loss = avg_real + avg_fake
        self.lg.scalar('discrim_real_loss_eval', avg_real, self.tr.step)
        self.lg.scalar('discrim_fake_loss_eval', avg_fake, self.tr.step)
        self.lg.scalar('discrim_loss_eval', avg_loss, self.tr.step)
        if avg_loss < self.tr.get_best():
            self.tr.set_best(avg_loss)
            self.verbose('Best validation loss : {:.4f} @ global step {}'.format(self.tr.get_best(), self.tr.step))
            sdist.save_atomic(self.discriminator.state_dict(), self.best_ckppath)
            self.verbose("Both the discriminator and ASR have been saved")

    def close(self):
        self.verbose("Finished training! The most recent model will" +
                     "be saved at step {} as well as the ASR model".format(self.tr.step))
        if self.train_step is not None:
            self.train_step.finish()
        if self.rank == 0:
            sdist.save_atomic(self.discriminator.state_dict(), self.ckppath)
            sdist.save_atomic(self.asr_model.state_dict(), self.asrpath_out)
        sdist.barrier()        # the next reader of these files (the Seed loop's next leg) runs on every rank


class SAETrainer(Solver):
    """Trains the speech autoencoder and, through it, the ASR model's Listener on audio alone
    (src/trainer.py:760-907; config 5's third leg).  With Adam (conf/default.yaml:24-26) one iteration of
    exec() is ONE engine.SAETrainStep call; any other optimizer type takes the reference's sequence (zero_grad,
    forward, backward, Solver.step over the speech autoencoder's parameters) with the torch optimizer.  The spectrogram figures of valid()
    (:871-887, matplotlib + librosa's specshow) are logged as the pair of arrays they would draw."""

    def __init__(self, config, paras):
        super().__init__(config, paras, 'sae')

    def load_data(self):
        n_jobs = self.set_if_exists('loader_jobs', 8)
        (self.mapper, _, self.train_set) = load_asr_dataset(
            self.config['sae']['train_index'], batch_size=self.train_batch_size, use_gpu=self.paras.gpu, n_jobs=n_jobs)
        (_, _, self.valid_set) = load_asr_dataset(
            self.config['sae']['valid_index'], batch_size=self.valid_batch_size, use_gpu=self.paras.gpu, n_jobs=n_jobs)

    def set_model(self, asrpath=None, asr_model=None):
        """src/trainer.py:774-796.  asr_model (this build): an ASR object already in memory."""
        from .engine import SAETrainStep
        from .speech_autoencoder import SpeechAutoEncoder
        (self.asrpath_in, self.asrpath_out) = self.genpath(asrpath, 'asr')
        self.asr_model = asr_model if asr_model is not None else self.setup_module(
            ASR, self.asrpath_in, self.mapper.get_dim(), **self.config['asr']['mdl'])
        self.speech_autoenc = self.setup_module(SpeechAutoEncoder, self.ckppath, self.asr_model.encoder.out_dim,
                                                self.config['asr']['mdl']['feature_dim'], **self.config['sae']['mdl'])
        opt = self.config['sae']['opt']
        self.train_step = None
        if opt['type'] == 'Adam' and self.device.type == 'cuda':
            self.train_step = SAETrainStep(self.asr_model, self.speech_autoenc, opt=(opt['type'], opt['learning_rate']),
                                           grad_clip=5.0)
            self.optim = self.train_step.optim
        else:
            # any other optimizer type: the reference's sequence (zero_grad, forward, backward, Solver.step over the
            # speech autoencoder's parameters) with the torch optimizer over both parameter lists (src/trainer.py:789-794)
            self.optim = getattr(torch.optim, opt['type'])(
                list(self.speech_autoenc.parameters()) + list(self.asr_model.encoder.parameters()),
                lr=opt['learning_rate'], eps=1e-8)

    def _forward_loss(self, x, x_lens):
        """(loss, prediction [B, 8 T', F]) of src/trainer.py:805-818."""
        if self.train_step is not None:
            return self.train_step.forward_loss(x, x_lens)
        from .seed_ops import sae_loss
        if not x.is_cuda:
            raise RuntimeError('ss_asr_amd computes on the GPU only (no CPU path)')
        listener_out, _ = self.asr_model.encoder(x, x_lens)
        pred = self.speech_autoenc(x, listener_out)
        return sae_loss(pred, x, max(x_lens)), pred

    def _rank_batches(self):
        from .gpu_loader import rank_batches
        mine = set(rank_batches(len(self.train_set), self.rank, self.world))
        for b_ind, (x, y) in enumerate(self.train_set):
            if b_ind in mine:
                yield b_ind, x

    def exec(self):
        self.verbose('Training set total {} batches.'.format(len(self.train_set)))
        nan_reported = self.train_step.skipped_steps if self.train_step is not None else 0
        epoch = 0
        while epoch < self.n_epochs:
            self.verbose("Starting epoch {} out of {}".format(epoch + 1, self.n_epochs))
            for b_ind, x in self._rank_batches():
                self.verbose('Batch: {}/{}, global step: {}'.format(b_ind, len(self.train_set), self.tr.step),
                             progress=True)
                x, x_lens = prepare_x(x, device=self.device)
                if self.train_step is not None:
                    loss = self.train_step(x, x_lens)
                    if self.train_step.skipped_steps > nan_reported:
                        nan_reported = self.train_step.skipped_steps
                        self.verbose('Error : grad norm is NaN @ step {}'.format(self.tr.step - 1))
                else:
                    self.optim.zero_grad()
                    loss, _ = self._forward_loss(x, x_lens)
                    loss.backward()
                    self.step(self.speech_autoenc.parameters(), self.optim)
                    if self.tr.step % self.logging_step == 0:
                        ops.check_persistent_status()
                if self.rank == 0 and self.tr.step % self.logging_step == 0:
                    self.lg.scalar('train_loss', loss.item(), self.tr.step)
                if self.tr.step % self.valid_step == 0:
                    self.valid()
                if self.rank == 0 and self.tr.step % self.save_step == 0:
                    if self.train_step is not None:
                        self.train_step.finish()       # never checkpoint after a time-out
                    self.verbose("Model saved at step {}".format(self.tr.step))
                    sdist.save_atomic(self.speech_autoenc.state_dict(), self.ckppath)
                    sdist.save_atomic(self.asr_model.state_dict(), self.asrpath_out)
                self.tr.do_step()
            epoch += 1
        if self.train_step is not None:
            self.train_step.finish()

    def valid(self):
        """src/trainer.py:840-897: the eval-mode loss (running batch-norm statistics) averaged over the
        validation batches; the last batch's utterances against their reconstructions go to the figure log."""
        self.speech_autoenc.eval()
        self.asr_model.eval()
        total, n_batches = 0.0, 0
        x = x_lens = pred = None
        with torch.no_grad():
            for b_idx, (x, y) in enumerate(self.valid_set):
                self.verbose('Validation step - {} ( {} / {} )'.format(self.tr.step, b_idx, len(self.valid_set)),
                             progress=True)
                x, x_lens = prepare_x(x, device=self.device)
                loss, pred = self._forward_loss(x, x_lens)
                total += float(loss)
                n_batches += 1
        ops.check_persistent_status()
        self.speech_autoenc.train()
        self.asr_model.train()
        if n_batches == 0 or self.rank != 0:
            return
        batch_t = max(x_lens)
        enc_final = torch.zeros(pred.shape[0], batch_t, pred.shape[2])
        enc_final[:, :pred.shape[1], :] = pred.cpu()
        for i in range(min(self.valid_batch_size, x.shape[0])):
            label_img = x[i, :x_lens[i], :].cpu().permute(1, 0)
            predict_img = enc_final[i, :x_lens[i], :].permute(1, 0)
            self.lg.figure('encode_compare_' + str(i), [label_img.numpy(), predict_img.numpy()], self.tr.step)
        avg_loss = total / n_batches
        self.lg.scalar('eval_loss', avg_loss, self.tr.step)
        if avg_loss < self.tr.get_best():
            self.tr.set_best(avg_loss)
            self.verbose('Best validation loss : {:.4f} @ global step {}'.format(self.tr.get_best(), self.tr.step))
            sdist.save_atomic(self.speech_autoenc.state_dict(), self.best_ckppath)
        else:
            self.verbose("Validation metric worse : ({:.4f} vs. {:.4f})".format(avg_loss, self.tr.get_best()))

    def close(self):
        self.verbose("Finished training! The most recent model will" +
                     "be saved at step {} as well as the ASR model".format(self.tr.step))
        if self.train_step is not None:
            self.train_step.finish()
        if self.rank == 0:
            sdist.save_atomic(self.speech_autoenc.state_dict(), self.ckppath)
            sdist.save_atomic(self.asr_model.state_dict(), self.asrpath_out)
        sdist.barrier()        # the next reader of these files (the Seed loop's next leg) runs on every rank


# src/train.py:19-20 offers the choice 'AdvTrainer' and resolves it with getattr(trainer, ...); the reference's
# trainer.py only defines ADVTrainer -- here the CLI's spelling resolves
AdvTrainer = ADVTrainer


def asr_seed_train(config, paras):
    """The Seed loop, src/trainer.py:1126-1177 (`train.py Seed`): super-iterations in which three trainers take
    turns on ONE ASR model, handed from leg to leg through checkpoints under <ckpdir>/<name>/ exactly as the
    reference chains them -- TAETrainer reads and writes asr_1.cpt (it trains the attention / speller half),
    ADVTrainer reads asr_1.cpt and the text autoencoder's checkpoint and writes asr_2.cpt (it trains the Listener
    against the text encoder), SAETrainer reads asr_2.cpt and writes asr_3.cpt (Listener again, through the speech
    autoencoder).  The count of super-iterations is `seed_train.its` (the key :1146 reads) or, when the config
    only has it, `seed_train.super_its` (the key conf/default.yaml:103-104 carries).

    `seed_train.legs` (this build; default ['tae', 'adv', 'sae'], the reference's sequence) may also name 'asr':
    a supervised ASRTrainer leg on the checkpoint the previous leg wrote."""
    ckpdir = os.path.join(paras.ckpdir, paras.name)
    seed = config['seed_train']
    its = seed['its'] if 'its' in seed else seed['super_its']
    legs = seed.get('legs', ['tae', 'adv', 'sae'])
    cpt = lambda k: os.path.join(ckpdir, 'asr_%d.cpt' % k)
    for i in range(its):
        print('Starting Super Iteration {}'.format(i + 1))
        tae_path, last = None, cpt(1)
        for leg in legs:
            if leg == 'tae':
                print('Starting TAE training')
                solver = TAETrainer(config, paras)
                solver.load_data()
                solver.set_model(asrpath=(cpt(1), cpt(1)))
                tae_path, last = solver.ckppath, cpt(1)
            elif leg == 'adv':
                print('Starting ADV training')
                solver = ADVTrainer(config, paras)
                solver.load_data()
                solver.set_model(taepath=tae_path, asrpath=(cpt(1), cpt(2)))
                last = cpt(2)
            elif leg == 'sae':
                print('Starting SAE training')
                solver = SAETrainer(config, paras)
                solver.load_data()
                solver.set_model(asrpath=(cpt(2), cpt(3)))
                last = cpt(3)
            elif leg == 'asr':
                print('Starting ASR training')
                solver = ASRTrainer(config, paras)
                solver.load_data()
                solver.set_model(asrpath=(last, last))
            else:
                raise ValueError("seed_train leg %r (known: 'tae', 'adv', 'sae', 'asr')" % (leg,))
            solver.exec()
            solver.close()
            del solver
