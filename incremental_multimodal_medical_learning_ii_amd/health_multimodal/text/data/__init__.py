from .io import TextInput, TypePrompts

__all__ = ["TextInput", "TypePrompts"]
