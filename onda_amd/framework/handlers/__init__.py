from .model_handler import get_model  # noqa: F401
from .adaptation_method_handler import get_adapt_method  # noqa: F401
