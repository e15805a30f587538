"""Drop-in for the reference's `transformer/evaluate.py`: `python -m asr_amd.evaluate -c config.yaml`.

Same surface (evaluate.py:18-103): `Evaluator().init_from_config(config)`, `beam_search(X) -> Y`,
`translate(src_path, output_path, batch_size)`, `evaluate(batch_size, **set)` (returns None), and the
same `uttid<TAB>hypothesis` output with BPE markers stripped.  Unlike the reference's __main__
(which overwrites `-c` with a hard-coded path, evaluate.py:120) the `-c` argument is honoured.
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

    def init_from_existed(self, model, data_reader):
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

    def translate(self, src_path, output_path, batch_size):
        logging.info('Translate %s.' % src_path)
        tmp = output_path + '.tmp'
        count, token_count = 0, 0
        start = time.time()
        meta = collections.deque()

        def features():
            for X, uttids in self.data_reader.get_test_batches(src_path, batch_size):
                meta.append((len(X), uttids))
                yield X
        with codecs.open(tmp, 'w', 'utf8') as fd:
            # == `Y = self.beam_search(X)` per batch, in order; with the build's `test.engines_per_gpu` >= 2 the next
            # batch is already being decoded while this one is written out
            for Y in self.model.predict_batches(features()):
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
    evaluator = Evaluator()
    evaluator.init_from_config(config)
    for attr in config.test:
        if attr.startswith('set'):
            evaluator.evaluate(config.test.batch_size, **config.test[attr])
    logging.info('Done')


if __name__ == '__main__':
    main()
