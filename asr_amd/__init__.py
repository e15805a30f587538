"""asr_amd -- MI355X (gfx950) implementation of the batched inference path of shiyuzh2007/ASR.

Importing the compute modules (`asr_amd.model`, `asr_amd.evaluate`) loads
`asr_amd/csrc/libasr_amd.so`; build it with `python -m asr_amd.build`.  The pure-host modules
(`asr_amd.utils`, `asr_amd.ark`, `asr_amd.weights`, `asr_amd.dist`) import without it.
"""
__all__ = ['utils', 'ark', 'weights', 'model', 'evaluate', 'dist', 'build']
