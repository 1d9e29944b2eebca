"""Consumers of an explicit solution (reference: src/ppopt/upop): device-backed point location and source-code export."""
from .linear_code_gen import generate_code_cpp, generate_code_js, generate_code_matlab
from .point_location import PointLocation
from .upop_payload import matlab_struct, payload_cpp, payload_js, save_matlab, upop_tables

__all__ = ['PointLocation', 'generate_code_cpp', 'generate_code_js', 'generate_code_matlab',
           'payload_cpp', 'payload_js', 'matlab_struct', 'save_matlab', 'upop_tables']
