"""N>1 path on CPU: world_size-2 gloo run of the sharding + hypothesis gather (asr_amd/dist.py).
The decode function is a deterministic stand-in (the HIP path needs a GPU); what is under test is
the bucketing, the deal of batches to ranks, and the gather restoring utterance order."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fake_decode(X):
    # hypothesis = function of the utterance's own content and length only
    n = X.shape[0]
    out = np.full((n, 6), 3, np.int32)
    for i in range(n):
        t = int((np.abs(X[i]).sum(-1) > 0).sum())
        out[i, :3] = [t % 50 + 4, int(abs(X[i, 0, 0]) * 1000) % 40 + 4, X.shape[1] % 30 + 4]
    return out


def _feats():
    rng = np.random.default_rng(0)
    lens = [55, 61, 120, 58, 49, 130, 131, 57, 300, 62, 63, 64]
    return [rng.standard_normal((t, 8)).astype(np.float32) for t in lens]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from asr_amd import dist as D
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    Y = D.decode_sharded(_fake_decode, _feats(), tokens_per_batch=100, beam=2, max_target_length=9, rank=rank,
                         world_size=world)
    q.put((rank, Y.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketing_and_assignment_cover_everything():
    sys.path.insert(0, ROOT)
    from asr_amd import dist as D
    lens = [len(f) for f in _feats()]
    batches = D.bucket_batches(lens, 100)
    assert sorted(sum(batches, [])) == list(range(len(lens)))
    for b in batches:                                            # one 10-frame bucket per batch
        assert len({(lens[i] - 50) // 10 if lens[i] >= 50 else -1 for i in b}) == 1
    costs = [D.batch_cost(len(b), max(lens[i] for i in b), 10, 200) for b in batches]
    own = D.assign_batches(costs, 3)
    assert sorted(sum(own, [])) == list(range(len(batches)))
    loads = [sum(costs[i] for i in o) for o in own]
    assert max(loads) < 2.0 * (sum(costs) / 3) + max(costs)


def test_world_size_2_gloo_matches_single_process():
    sys.path.insert(0, ROOT)
    from asr_amd import dist as D
    single = D.decode_sharded(_fake_decode, _feats(), tokens_per_batch=100, beam=2, max_target_length=9)
    assert single.shape == (12, 10)
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] == res[1] == single.tolist()


def test_decode_sharded_with_several_decode_fns_gives_the_same_rows_and_propagates_errors():
    """decode_fn may be a list (one callable per engine handle of the rank's GPU): the batches are decoded by that many
    host threads; same result as one callable; an exception in a worker is re-raised on the caller."""
    import threading
    import time
    from asr_amd import dist as D
    rng = np.random.default_rng(5)
    feats = [rng.standard_normal((int(n), 4)).astype(np.float32) for n in rng.integers(50, 140, 40)]
    one = D.decode_sharded(_fake_decode, feats, 300, 2, 6)
    used = set()

    def tagged(X):
        used.add(threading.get_ident())
        time.sleep(0.01)                                     # long enough for the other threads to get their turn
        return _fake_decode(X)
    many = D.decode_sharded([tagged, tagged, tagged], feats, 300, 2, 6)
    assert np.array_equal(one, many)
    assert len(used) >= 2

    def boom(X):
        raise RuntimeError('decode failed')
    with pytest.raises(RuntimeError, match='decode failed'):
        D.decode_sharded([boom, boom], feats, 300, 2, 6)


def test_merge_jobs_and_merged_decode_sharded():
    """dist.merge_jobs joins a rank's batches, in order of padded length, into jobs of at most `merge_utts` utterances (a batch
    larger than that stays a job of its own); decode_sharded then hands decode_fn LISTS of padded batches and takes back the
    list of their Y - same gathered hypotheses as batch by batch."""
    from asr_amd import dist as D
    sizes = [3, 1, 9, 2, 2, 4]
    tmax = [50, 20, 30, 20, 90, 31]
    assert D.merge_jobs([0, 1, 2, 3, 4, 5], sizes, tmax, 0) == [[0], [1], [2], [3], [4], [5]]
    jobs = D.merge_jobs([0, 1, 2, 3, 4, 5], sizes, tmax, 6)
    assert jobs == [[1, 3], [2], [5], [0, 4]]                       # by length: 1, 3 (20) | 2 (30, nine utterances alone) | 5 (31) | 0, 4
    assert sorted(sum(jobs, [])) == [0, 1, 2, 3, 4, 5]
    assert D.merge_jobs([4, 0], sizes, tmax, 100) == [[0, 4]]

    rng = np.random.default_rng(3)
    lens = [int(x) for x in rng.integers(50, 140, size=23)]
    feats = [np.full((ln, 4), float(i + 1), np.float32) for i, ln in enumerate(lens)]
    calls = []

    def fake(X):                                                     # hypothesis = [utterance id, its padded length]
        return np.stack([np.array([int(X[r, 0, 0]) - 1, X.shape[1]], np.int32) for r in range(X.shape[0])])

    def fake_groups(Xs):
        calls.append([X.shape[0] for X in Xs])
        return [fake(X) for X in Xs]

    want = D.decode_sharded(fake, feats, 300, 2, 7)
    got = D.decode_sharded(fake_groups, feats, 300, 2, 7, merge_utts=6)
    assert np.array_equal(got, want)
    assert all(sum(c) <= 6 or len(c) == 1 for c in calls) and any(len(c) > 1 for c in calls)
    assert np.array_equal(D.decode_sharded([fake_groups, fake_groups], feats, 300, 2, 7, merge_utts=6), want)
    with pytest.raises(RuntimeError):
        D.decode_sharded(lambda Xs: [], feats, 300, 2, 7, merge_utts=6)


# ---- the drop-in surface under a launcher: Evaluator.translate with WORLD_SIZE > 1 ---------------------------------------
class _FakeModel(object):
    """Stands in for asr_amd.model.Model where there is no GPU: a hypothesis depends on the utterance's own frames AND on
    how far its batch is padded (like the real path: Q1 / Q7), so a rank that re-batched would be caught."""

    def __init__(self, config):
        self._config = config
        self.calls = []

    def predict_batches(self, batches):
        for X in batches:
            self.calls.append(X.shape)
            yield self.predict(X)

    def predict(self, X):
        n = X.shape[0]
        out = np.full((n, 5), 3, np.int32)
        for i in range(n):
            t = int((np.abs(X[i]).sum(-1) > 0).sum())
            out[i, :4] = [4 + t % 9, 4 + int(abs(X[i, 0, 0]) * 1000) % 9, 4 + X.shape[1] % 9, 4 + (i % 2) * 3]
            if t % 4 == 0:
                out[i, 2:] = 3                               # some hypotheses end early
        return out


def _write_test_set(tmp, n_utts=11):
    from asr_amd import ark
    rng = np.random.default_rng(12)
    lens = [int(x) for x in rng.integers(50, 140, size=n_utts)]
    scp, arkf = os.path.join(tmp, 'feats.scp'), os.path.join(tmp, 'feats.ark')
    w = ark.ArkWriter(scp)
    for i, ln in enumerate(lens):
        w.write_next_utt(arkf, 'utt%02d' % i, rng.standard_normal((ln, 8)).astype(np.float32))
    w.close()
    vocab = os.path.join(tmp, 'vocab.txt')
    with open(vocab, 'w') as f:
        for i, wd in enumerate(['<PAD>', '<UNK>', '<S>', '</S>'] + [('t%d@@' % i if i % 4 == 0 else 't%d' % i) for i in range(12)]):
            f.write('%s %d\n' % (wd, 1000 - i))
    return scp, vocab


def _translate_config(scp, vocab, out_path, bucketed):
    from asr_amd.utils import AttrDict
    test = {'batch_size': 3, 'max_target_length': 8, 'lp_alpha': 0.6, 'beam_size': 2, 'num_gpus': 1,
            'set1': {'src_path': scp, 'output_path': out_path}}
    if bucketed:
        test.update({'bucketed_reader': True, 'tokens_per_batch': 150})
    return AttrDict({'model': 'Transformer', 'dst_vocab': vocab, 'dst_vocab_size': 16, 'train': {'input_dim': 8}, 'test': test})


def _translate_worker(rank, world, port, scp, vocab, out_path, bucketed, q):
    sys.path.insert(0, ROOT)
    os.environ.update({'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port), 'RANK': str(rank), 'WORLD_SIZE': str(world),
                       'LOCAL_RANK': str(rank), 'ASR_AMD_DIST_BACKEND': 'gloo'})
    import torch.distributed as dist
    from asr_amd.evaluate import Evaluator
    from asr_amd.utils import DataReader
    cfg = _translate_config(scp, vocab, out_path, bucketed)
    model = _FakeModel(cfg)
    ev = Evaluator()
    ev.init_from_existed(model, None, DataReader(cfg))                 # the reference's three-argument form (evaluate.py:40)
    assert ev.evaluate(cfg.test.batch_size, **cfg.test.set1) is None
    q.put((rank, [tuple(s) for s in model.calls]))
    dist.destroy_process_group()


@pytest.mark.parametrize('bucketed', [False, True])
def test_translate_under_two_ranks_writes_the_single_process_file(tmp_path, bucketed):
    """WORLD_SIZE = 2 (gloo): every rank forms the same batches, decodes its share, one all_gather, rank 0 writes - the
    bytes of the single-process run; scp-order batches (evaluate.py:59) and the bucketed reader (utils.py:285-323)."""
    sys.path.insert(0, ROOT)
    from asr_amd.evaluate import Evaluator
    from asr_amd.utils import DataReader
    scp, vocab = _write_test_set(str(tmp_path))
    single = str(tmp_path / 'single.txt')
    cfg = _translate_config(scp, vocab, single, bucketed)
    m = _FakeModel(cfg)
    ev = Evaluator()
    ev.init_from_existed(m, DataReader(cfg))                           # (the two-argument form of earlier rounds still works)
    ev.translate(scp, single, cfg.test.batch_size)
    want = open(single, 'rb').read()
    assert want.count(b'\n') == 11 and b'@@' not in want
    sharded = str(tmp_path / 'sharded.txt')
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_translate_worker, args=(r, 2, port, scp, vocab, sharded, bucketed, q)) for r in range(2)]
    for p in procs:
        p.start()
    calls = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert open(sharded, 'rb').read() == want
    assert not os.path.exists(sharded + '.tmp')
    # the two ranks decoded disjoint shares that together are exactly the single process's batches
    assert sorted(calls[0] + calls[1]) == sorted(tuple(s) for s in m.calls)
    assert calls[0] and calls[1]


def test_predict_batches_sharded_single_process_and_errors():
    from asr_amd import dist as D
    rng = np.random.default_rng(1)
    batches = [rng.standard_normal((n, t, 4)).astype(np.float32) for n, t in [(3, 60), (1, 90), (4, 55), (2, 120)]]
    m = _FakeModel(None)
    Ys = D.predict_batches_sharded(list(batches), m.predict_batches, beam=2, max_target_length=7)
    assert [Y.shape for Y in Ys] == [(3, 8), (1, 8), (4, 8), (2, 8)]
    for X, Y in zip(batches, Ys):
        assert np.array_equal(Y[:, :5], m.predict(X)) and np.all(Y[:, 5:] == 3)
    with pytest.raises(RuntimeError, match='ended early'):
        D.predict_batches_sharded(list(batches), lambda it: iter([m.predict(batches[0])]), 2, 7)
    with pytest.raises(ValueError, match='only its shape'):
        D.predict_batches_sharded([(3, 60)] + batches[1:], m.predict_batches, 2, 7)
