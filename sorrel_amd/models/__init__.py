from sorrel_amd.models.base_model import BaseModel, RandomModel

__all__ = ["BaseModel", "RandomModel"]
