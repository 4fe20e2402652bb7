"""cadence-rag_amd — MI355X-native dense-retrieval lane for Cadence RAG.

Drop-in counterpart of the reference's dense path only (app/embeddings.py,
app/embedding_pipeline.py, the dense lane of app/retrieve.py).  See DESIGN.md.
"""
__version__ = "0.1.0"
