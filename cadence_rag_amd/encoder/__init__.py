"""In-process Qwen3-Embedding encoder for MI355X: hand-written HIP operators (csrc/crag_encoder.hip)
+ library GEMMs, packed variable-length batches, last-token pooling -> [:1024] -> L2 normalise —
the math the reference delegates to its external Triton/ONNX gateway
(/root/reference/P620_TRITON_QWEN3_4B_EMBEDDING_RUNBOOK.md:683-716)."""
from .qwen3 import PackedBatch, Qwen3Config, Qwen3Encoder  # noqa: F401
