"""Drop-in for the reference's `transformer/evaluate.py`: `python -m asr_amd.evaluate -c config.yaml`.

Same surface (evaluate.py:18-103): `Evaluator().init_from_config(config)`, `beam_search(X) -> Y`,
`translate(src_path, output_path, batch_size)`, `evaluate(batch_size, **set)` (returns None), and the
same `uttid<TAB>hypothesis` output with BPE markers stripped.  Unlike the reference's __main__
(which overwrites `-c` with a hard-coded path, evaluate.py:120) the `-c` argument is honoured.

Multi-GPU: the reference replicates towers inside one graph (model.py:168-201); here it is one process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m asr_amd.evaluate -c config.yaml

Every rank forms the same batches, decodes its share on the GPU LOCAL_RANK names, one RCCL all_gather collects the
hypotheses and rank 0 writes the file - the same bytes as the single-process run (`translate`, asr_amd/dist.py).
"""
from __future__ import annotations

import codecs
import collections
import logging
import os
import re
import time
from argparse import ArgumentParser

import numpy as np
import yaml

from . import weights
from .model import Model, Transformer   # noqa: F401  (config.model names the class, evaluate.py:26)
from .utils import AttrDict, DataReader

_MODELS = {'Transformer': Transformer}
_BPE = re.compile(r'(@@ )|(@@ ?$)')


class Evaluator(object):
    """Evaluate the model (evaluate.py:18-103)."""

    def __init__(self):
        pass

    def init_from_config(self, config, variables=None):
        """Build the test model on `config.test.num_gpus` towers and restore the latest checkpoint of
        `config.model_dir` (evaluate.py:25-38).  `variables` (name -> array) bypasses the checkpoint."""
        if config.model not in _MODELS:
            raise ValueError('unknown model %r' % config.model)
        self.model = _MODELS[config.model](config, config.test.num_gpus)
        self.model.build_test_model()
        if variables is None:
            variables = weights.load_checkpoint(weights.latest_checkpoint(config.model_dir))
        self.model.restore(variables)
        self.data_reader = DataReader(config)

    def init_from_existed(self, model, sess, data_reader=None):
        """evaluate.py:40-44.  `sess` (the reference's tf.Session) has no counterpart here and is ignored; the two-argument
        form `init_from_existed(model, data_reader)` of this build's earlier rounds still works."""
        if data_reader is None and hasattr(sess, 'indices_to_words'):
            sess, data_reader = None, sess
        self.sess = sess
        self.model = model
        self.data_reader = data_reader

    def beam_search(self, X):
        """X float32 [N, Tmax, input_dim] -> int32 [N, Lmax] padded with 3 (evaluate.py:46-47)."""
        return self.model.predict(X)

    def loss(self, X, Y):
        """Sum over target tokens of the cross entropy of the teacher-forced decoder (evaluate.py:49-50)."""
        return self.model.loss(X, Y)

    def ppl(self, src_path, dst_path, batch_size):
        """Perplexity of the targets in `dst_path` (evaluate.py:76-87)."""
        logging.info('Calculate PPL for %s and %s.' % (src_path, dst_path))
        token_count = 0
        loss_sum = 0
        for batch in self.data_reader.get_test_batches_with_target(src_path, dst_path, batch_size):
            X, Y = batch
            loss_sum += self.loss(X, Y)
            token_count += np.sum(np.greater(Y, 0))
        ppl = np.exp(loss_sum / token_count)
        logging.info('PPL: %.4f' % ppl)
        return ppl

    def _test_config(self):
        for holder in (getattr(self, 'data_reader', None), getattr(self, 'model', None)):
            cfg = getattr(holder, '_config', None)
            if cfg is not None and cfg.test is not None:
                return cfg.test
        return None

    def _test_batches(self, src_path, batch_size):
        """The reader `translate` batches with: scp order, `batch_size` utterances (evaluate.py:59), or - build extension
        `test.bucketed_reader: true` - the reference's length-bucketed reader with `test.tokens_per_batch` (utils.py:285-323),
        which north_star's multi-GPU sharding names.  Either way the SAME batches in the same order for any number of ranks."""
        t = self._test_config()
        if t is not None and t.bucketed_reader:
            return self.data_reader.get_test_batches_with_buckets(src_path, int(t.tokens_per_batch or 7000))
        return self.data_reader.get_test_batches(src_path, batch_size)

    def _predict_batches(self, batches):
        """`Y = self.beam_search(X)` for every X, in order: through the model's `predict_batches` when it has one (several
        batches in flight per GPU), else batch by batch through `beam_search` (a model that only implements `predict`, a
        subclass that overrides `beam_search`: evaluate.py:46-66 as written)."""
        pb = getattr(self.model, 'predict_batches', None)
        if pb is not None and type(self).beam_search is Evaluator.beam_search:
            return pb(batches)
        return (self.beam_search(X) for X in batches)

    def translate(self, src_path, output_path, batch_size):
        logging.info('Translate %s.' % src_path)
        from . import dist as D
        world, rank = D.world()
        tmp = output_path + '.tmp'
        count, token_count = 0, 0
        start = time.time()
        meta = collections.deque()
        if world > 1:
            # One process per GPU (python -m torch.distributed.run ... -m asr_amd.evaluate -c cfg): every rank forms the same
            # batches, whole batches are dealt over the ranks by cost, one all_gather of the hypotheses, rank 0 writes the
            # file - byte for byte the single-process file (= the towers' pad-with-3 + concat, model.py:192-201).
            device = D.init_from_env()
            batches = []
            for X, uttids in self._test_batches(src_path, batch_size):
                meta.append((len(X), uttids))
                batches.append(X)
            t = self._test_config()
            if t is None:
                raise ValueError('multi-process translate needs the test configuration (beam_size, max_target_length)')
            owners = D.assign_batches([D.batch_cost(X.shape[0], X.shape[1], int(t.beam_size), int(t.max_target_length)) for X in batches], world)
            mine = set(owners[rank])
            batches = [X if i in mine else (X.shape[0], X.shape[1]) for i, X in enumerate(batches)]      # keep only this rank's features
            results = D.predict_batches_sharded(batches, self._predict_batches, int(t.beam_size), int(t.max_target_length),
                                                rank=rank, world_size=world, device=device)
        else:
            def features():
                for X, uttids in self._test_batches(src_path, batch_size):
                    meta.append((len(X), uttids))
                    yield X
            # == `Y = self.beam_search(X)` per batch, in order; with several engine handles per GPU (`test.engines_per_gpu`) the
            # next batches are already being decoded while this one is written out
            results = self._predict_batches(features())
        if rank == 0:
            with codecs.open(tmp, 'w', 'utf8') as fd:
                for Y in results:
                    n_x, uttids = meta.popleft()
                    sents = self.data_reader.indices_to_words(Y)
                    assert n_x == len(sents)
                    for sent, uttid in zip(sents, uttids):
                        fd.write(uttid + '\t' + sent + '\n')
                    count += n_x
                    token_count += int(np.sum(np.not_equal(Y, 3)))   # 3: </s>
                    time_span = time.time() - start
                    logging.info('{0} sentences ({1} tokens) processed in {2:.2f} minutes (speed: {3:.4f} sec/token).'.
                                 format(count, token_count, time_span / 60, time_span / max(token_count, 1)))
            # Remove BPE flag, if have (the reference shells out to: sed -r 's/(@@ )|(@@ ?$)//g').
            with codecs.open(tmp, 'r', 'utf8') as fi, codecs.open(output_path, 'w', 'utf8') as fo:
                for line in fi:
                    fo.write(_BPE.sub('', line.rstrip('\n')) + '\n')
            os.remove(tmp)
            logging.info('The result file was saved in %s.' % output_path)
        if world > 1:
            import torch.distributed as dist
            dist.barrier()                                           # nobody returns before rank 0's file is complete

    def evaluate(self, batch_size, **kargs):
        """Evaluate the model on dev set (evaluate.py:89-103; BLEU/PPL hooks are commented out there)."""
        self.translate(kargs['src_path'], kargs['output_path'], batch_size)
        return None


def load_config(path):
    with open(path) as f:
        return AttrDict(yaml.safe_load(f))


def main(argv=None):
    parser = ArgumentParser()
    parser.add_argument('-c', '--config', dest='config', required=True)
    args = parser.parse_args(argv)
    config = load_config(args.config)
    logging.basicConfig(level=logging.INFO)
    from . import dist as D
    world, _ = D.world()
    if world > 1:
        D.init_from_env()             # one process per GPU: this rank decodes on the GPU LOCAL_RANK names (asr_amd.model.default_device)
    evaluator = Evaluator()
    evaluator.init_from_config(config)
    for attr in config.test:
        if attr.startswith('set'):
            evaluator.evaluate(config.test.batch_size, **config.test[attr])
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    logging.info('Done')


if __name__ == '__main__':
    main()
