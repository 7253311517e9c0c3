"""Visualisation helpers of the reference (`health_multimodal/common/visualization.py`) are out of the hot path."""
