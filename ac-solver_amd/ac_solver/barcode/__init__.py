"""Neighbourhood sizes in the Andrews-Curtis graph -- the workload of the reference's C++ side program
barcode_analysis/5_steps_neibourhoods, on the GPU."""
from ac_solver.barcode.neighbourhoods import neighbourhood_sizes, neighbourhood_sizes_of_file
from ac_solver.barcode.simplex_data import simplex_graph, write_simplex_files

__all__ = ["neighbourhood_sizes", "neighbourhood_sizes_of_file", "simplex_graph", "write_simplex_files"]
