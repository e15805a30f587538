"""Writes the inference configs shipped in configs/ (same YAML schema and key names as the reference's
transformer/config_template_*.yaml; values of the inference-relevant keys taken from those files,
dataset paths replaced by placeholders).  The reference's own YAML files are accepted unchanged too."""
import os

UNITS = [  # name, dst_vocab_size, test.batch_size, test.num_gpus, extra test keys
    ('char', 3900, 100, 4, ''),
    ('pinyin', 1388, 100, 4, ''),
    ('word', 28444, 100, 4, ''),
    ('bpe', 11039, 100, 4, ''),
    ('ciphone', 122, 100, 4, ''),
    ('char_seame', 3205, 50, 8, '    tokens_per_batch: 7000\n'),
]
TEMPLATE = """---
model: 'Transformer'
dst_vocab: './data/{name}/vocab.txt'
dst_vocab_size: {vocab}
hidden_units: 1024
scale_embedding: True
tie_embedding_and_softmax: True
attention_dropout_rate: 0.0
residual_dropout_rate: 0.3
num_blocks: 6
num_heads: 16
ff_activation: 'glu'
model_dir: './exp/{name}'
train:
    input_dim: 320
test:
    batch_size: {batch}
{extra}    max_target_length: 200
    lp_alpha: 0.6
    beam_size: 13
    num_gpus: {gpus}

    set1:
        src_path: './data/{name}/dev/feats.scp.left3_sub3'
        ref_path: './data/{name}/dev/text'
        output_path: './exp/{name}/decoder.txt'
        cmd:
"""
here = os.path.dirname(os.path.abspath(__file__))
for name, vocab, batch, gpus, extra in UNITS:
    with open(os.path.join(here, 'asr_%s_big_left3_dim80.yaml' % name), 'w') as f:
        f.write(TEMPLATE.format(name=name, vocab=vocab, batch=batch, gpus=gpus, extra=extra))
print('ok')
