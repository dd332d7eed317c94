from .retrieval import (BaseVisualRetrieverProcessor, CustomRetrievalEvaluator, get_torch_device,  # noqa: F401
                        score_multi_vector_masked)
