"""Consumers of an explicit solution (reference: src/ppopt/upop): device-backed point location and source-code export."""
from .linear_code_gen import generate_code_cpp, generate_code_js, generate_code_matlab
from .point_location import PointLocation

__all__ = ['PointLocation', 'generate_code_cpp', 'generate_code_js', 'generate_code_matlab']
