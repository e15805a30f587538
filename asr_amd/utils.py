"""Host-side harness of the inference path: config access, test batching, hypothesis -> text.

Mirrors the parts of the reference's `transformer/utils.py` that `evaluate.py` uses
(AttrDict :27-40, DataReader :43-73,204-220,285-323,384-417, expand_feed_dict :420-443), with the
same names, argument meaning and results.  Training-side batching is out of scope.
"""
from __future__ import annotations

import codecs
import logging

import numpy as np

from . import ark as zark

PAD_INDEX = 0
UNK_INDEX = 1
BOS_INDEX = 2
EOS_INDEX = 3

PAD = u'<PAD>'
UNK = u'<UNK>'
BOS = u'<S>'
EOS = u'</S>'


class AttrDict(dict):
    """Dictionary whose keys can be read as attributes; a missing key reads as None and a nested
    plain dict is wrapped on first access (utils.py:27-40)."""

    def __getattr__(self, item):
        if item.startswith('__') and item.endswith('__'):      # copy / pickle protocol probes are not configuration keys
            raise AttributeError(item)
        if item not in self:
            return None
        if type(self[item]) is dict:
            self[item] = AttrDict(self[item])
        return self[item]


class DataReader(object):
    """Vocabulary + test-time batching (utils.py:43-73, 204-220, 285-323, 384-417)."""

    def __init__(self, config):
        self._config = config
        self.load_vocab()

    def load_vocab(self):
        """First column of the first `dst_vocab_size` lines; entries 0..3 are <PAD> <UNK> <S> </S>."""
        path, vocab_size = self._config.dst_vocab, self._config.dst_vocab_size
        logging.info('Load vocabularies %s.' % path)
        with codecs.open(path, 'r', 'utf-8') as f:
            vocab = [line.split()[0] for line in f if line.split()]
        vocab = vocab[:vocab_size]
        assert len(vocab) == vocab_size, 'vocab file has %d entries, config wants %d' % (len(vocab), vocab_size)
        self.dst2idx = {w: i for i, w in enumerate(vocab)}
        self.idx2dst = {i: w for i, w in enumerate(vocab)}

    @staticmethod
    def _create_feat_batch(feats):
        """Zero-pad [T_u, D] utterances to the batch maximum -> ([N, Tmax, D] float32, mask [N, Tmax])
        with mask 0 on real frames and 1 on padding (utils.py:204-220)."""
        maxlen = max(len(f) for f in feats)
        dim = np.shape(feats[0])[1]
        batch = np.zeros([len(feats), maxlen, dim], dtype=np.float32)
        mask = np.ones([len(feats), maxlen], dtype=np.int32)
        for i, f in enumerate(feats):
            batch[i, :len(f)] = f
            mask[i, :len(f)] = 0
        return batch, mask

    @staticmethod
    def _create_target_batch(sents, phone2idx):
        """Token lists -> ids with </S> appended, unknown words -> <UNK> (1), zero-padded to the batch maximum:
        ([N, Lmax] int32, mask) (utils.py:222-250)."""
        indices = [[phone2idx.get(w, 1) for w in (list(sent) + [u'</S>'])] for sent in sents]
        maxlen = max(len(x) for x in indices)
        batch = np.zeros([len(sents), maxlen], np.int32)
        mask = np.ones([len(sents), maxlen], dtype=np.int32)
        for i, x in enumerate(indices):
            batch[i, :len(x)] = x
            mask[i, :len(x)] = 0
        return batch, mask

    def get_test_batches_with_target(self, src_path, dst_path, batch_size):
        """(X, Y) batches for Evaluator.ppl (evaluate.py:76-87).  The reference calls this reader but does not
        define it; this is the composition of its own pieces: the `uttid w1 w2 ...` target map of the training
        reader (utils.py:108-116), scp order and batch size of get_test_batches (:384-403), targets through
        _create_target_batch (:222-250).  Utterances without a target line are skipped with a warning."""
        targets = {}
        with codecs.open(dst_path, 'r', 'utf-8') as f:
            for line in f:
                parts = line.split()
                if parts:
                    targets[parts[0]] = parts[1:]
        reader = zark.ArkReader(src_path)
        feats, sents = [], []
        while True:
            uttid, feat, looped = reader.read_next_utt()
            if looped:
                break
            if uttid not in targets:
                logging.warning('uttid=%s has no target, skipped' % uttid)
                continue
            feats.append(feat)
            sents.append(targets[uttid])
            if len(feats) >= batch_size:
                yield self._create_feat_batch(feats)[0], self._create_target_batch(sents, self.dst2idx)[0]
                feats, sents = [], []
        if feats:
            yield self._create_feat_batch(feats)[0], self._create_target_batch(sents, self.dst2idx)[0]

    def get_test_batches(self, src_path, batch_size):
        """scp order, `batch_size` utterances per batch, no bucketing (utils.py:384-403)."""
        reader = zark.ArkReader(src_path)
        cache, uttids = [], []
        while True:
            uttid, feat, looped = reader.read_next_utt()
            if looped:
                break
            cache.append(feat)
            uttids.append(uttid)
            if len(cache) >= batch_size:
                yield self._create_feat_batch(cache)[0], uttids
                cache, uttids = [], []
        if cache:
            yield self._create_feat_batch(cache)[0], uttids

    def get_test_batches_with_buckets(self, src_path, tokens_per_batch):
        """Length-bucketed batches: buckets 50, 60, ... 9990 on the stacked length; a bucket is emitted
        once its frame count exceeds `tokens_per_batch`; leftovers in bucket order (utils.py:285-323)."""
        buckets = list(range(50, 10000, 10))

        def select_bucket(sl):
            for b in buckets:
                if sl < b:
                    return b
            raise Exception('The sequence is too long: ({})'.format(sl))

        caches = {b: [[], [], 0] for b in buckets}
        reader = zark.ArkReader(src_path)
        count = 0
        while True:
            uttid, feat, looped = reader.read_next_utt()
            if looped:
                break
            b = select_bucket(len(feat))
            caches[b][0].append(feat)
            caches[b][1].append(uttid)
            caches[b][2] += len(feat)
            count += 1
            if caches[b][2] > tokens_per_batch:
                yield self._create_feat_batch(caches[b][0])[0], caches[b][1]
                caches[b] = [[], [], 0]
        for b in buckets:
            if caches[b][0]:
                yield self._create_feat_batch(caches[b][0])[0], caches[b][1]
        logging.info('get_test_batches_with_buckets, loaded count=%d' % count)

    def indices_to_words(self, Y, o='dst'):
        """ids -> tokens up to the first </S> (3), joined by spaces (utils.py:405-417)."""
        assert o == 'dst'
        sents = []
        for y in Y:
            sent = []
            for i in y:
                if i == 3:
                    break
                sent.append(self.idx2dst[int(i)])
            sents.append(' '.join(sent))
        return sents


def shard_bounds(batch_size, n):
    """Dim-0 split used by expand_feed_dict: the first (batch_size % n) shards get one extra row."""
    span, remainder = batch_size // n, batch_size % n
    bounds, base = [], 0
    for i in range(n):
        end = base + span + (1 if i < remainder else 0)
        bounds.append((base, end))
        base = end
    return bounds


def expand_feed_dict(feed_dict):
    """If a key is a tuple of placeholders, split its value along dim 0 over them (utils.py:420-443)."""
    new_feed_dict = {}
    for k, v in feed_dict.items():
        if type(k) is not tuple:
            new_feed_dict[k] = v
        else:
            for p, (a, b) in zip(k, shard_bounds(v.shape[0], len(k))):
                new_feed_dict[p] = v[a:b]
    return new_feed_dict
