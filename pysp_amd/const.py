"""Demosaic quality levels and CFA family, value-compatible with the reference's enums (const.py:3-8):
Draft = 1, Fast = 2, Best = 3; Rgbg = 1."""
import enum

QualityDemosaic = enum.Enum("QualityDemosaic", ("Draft", "Fast", "Best"), module=__name__)
QualityDemosaic.__doc__ = "Draft: aligned quarter-resolution resize; Fast: edge-assisted Gaussian; Best: AHD."

PatternDemosaic = enum.Enum("PatternDemosaic", ("Rgbg",), module=__name__)
