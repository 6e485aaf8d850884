"""Data formats that feed the path (reference: rrmpg/data/__init__.py)."""

from .camelsloader import CAMELSLoader  # noqa: F401
