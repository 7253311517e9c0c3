"""MI355X-native mirror of the reference's vendored `health_multimodal` package (hi-ml-multimodal 0.1.3,
`health_multimodal/__init__.py:6`): same module paths, class names, method signatures and state-dict keys;
the arithmetic runs on the cxrk HIP kernels."""
__version__ = "0.1.3+cxrk"
