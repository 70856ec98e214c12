# Builds oracle/_ref/ from the reference's own sources where they lie (only when /root/reference is present; the GPU box
# uses the prebuilt file).  Nothing of the reference is copied into the repository; oracle/_ref/ is git-ignored.
REF ?= /root/reference
all: _ref/libspline_ref.so

_ref/libspline_ref.so: ref_spline.cpp $(REF)/headers/spline.h
	@mkdir -p _ref
	g++ -O2 -std=c++11 -fPIC -shared -I$(REF)/headers ref_spline.cpp -o $@
