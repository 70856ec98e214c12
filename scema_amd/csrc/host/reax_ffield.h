// reax_ffield.h -- reader of ReaxFF force-field files for the product path (what `pair_coeff * * ffield.reax.2 H C N O` does in
// lammps_scripts_reax/in.strain.lammps:11): the tables of the elements named, in the layout the kernels use (reax/rx_types.h).
#pragma once
#include <string>
#include <vector>

#include "../reax/rx_types.h"

namespace scema {

// elements[k] = element symbol of LAMMPS atom type k + 1.  On success P holds the tables of the distinct elements (compact
// type index = order of first appearance) and type_map[k] the compact index of LAMMPS type k + 1.
bool read_reax_ffield(const std::string &path, const std::vector<std::string> &elements, RxParams &P, std::vector<int> &type_map, std::string &err);

}  // namespace scema
