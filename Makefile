# MI355X (gfx950) BWT block codec -- same targets as the reference's Makefile
# (all / test / performance / clean / rebuild; reference Makefile:52-55,173-186,256-266)
# plus `lib` for the HIP library.  The C++ host layer is plain C++17 (g++); only
# libbwtgpu.so needs hipcc.  Objects that belong to test binaries are compiled with
# -DBUILD_TESTS, which removes the CLIs' main() as in the reference (Makefile:39-45).
CXX      := g++
HIPCC    ?= hipcc
CXXFLAGS := -std=c++17 -Wall -Wextra -O3
BUILD    := build
HOST     := bwt_amd/host
CSRC     := bwt_amd/csrc
TESTS    := tests/cpp
LIB      := bwt_amd/libbwtgpu.so
INCLUDES := -Iinclude -I$(HOST)
LDFLAGS  := -Lbwt_amd -lbwtgpu -Wl,-rpath,'$$ORIGIN/../bwt_amd' -pthread

MAIN_EXECS := $(BUILD)/bwt $(BUILD)/inverse_bwt
TEST_EXECS := $(BUILD)/test_small $(BUILD)/test_medium $(BUILD)/test_bzip2_format
PERF_EXECS := $(BUILD)/performance $(BUILD)/compare_bwt_performance
ORIGIN_EXECS := $(BUILD)/bzip2_bwt_extractor $(BUILD)/bzip2_inverse_bwt_extractor
TEST_LIB_OBJS := $(BUILD)/src_test/bwt.o $(BUILD)/src_test/inverse_bwt.o $(BUILD)/src_test/file_processor.o \
                 $(BUILD)/src_test/bzip2_format.o
HOST_HDRS := $(wildcard $(HOST)/*.hpp) include/bwtgpu.h

.PHONY: all lib test performance clean rebuild
all: lib $(MAIN_EXECS) $(TEST_EXECS) $(PERF_EXECS) $(ORIGIN_EXECS)
lib: $(LIB)

$(LIB): $(CSRC)/bwtgpu.hip $(wildcard $(CSRC)/*.hpp) include/bwtgpu.h
	@echo "Building $@ (gfx950)"
	@$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude -I$(CSRC) $(CSRC)/bwtgpu.hip -o $@

$(BUILD)/bwt: $(BUILD)/src/bwt.o $(BUILD)/src/file_processor.o | $(LIB)
	@echo "Building $@"
	@$(CXX) $(CXXFLAGS) $^ -o $@ $(LDFLAGS)
$(BUILD)/inverse_bwt: $(BUILD)/src/inverse_bwt.o $(BUILD)/src/file_processor.o | $(LIB)
	@echo "Building $@"
	@$(CXX) $(CXXFLAGS) $^ -o $@ $(LDFLAGS)

$(TEST_EXECS) $(PERF_EXECS) $(ORIGIN_EXECS): $(BUILD)/%: $(TESTS)/%.cpp $(TEST_LIB_OBJS) $(HOST_HDRS) $(wildcard $(TESTS)/*.hpp) | $(LIB)
	@echo "Building $@"
	@$(CXX) $(CXXFLAGS) -DBUILD_TESTS $(INCLUDES) $< $(TEST_LIB_OBJS) -o $@ $(LDFLAGS)

$(BUILD)/src/%.o: $(HOST)/%.cpp $(HOST_HDRS)
	@mkdir -p $(dir $@)
	@$(CXX) $(CXXFLAGS) $(INCLUDES) -c $< -o $@
$(BUILD)/src_test/%.o: $(HOST)/%.cpp $(HOST_HDRS)
	@mkdir -p $(dir $@)
	@$(CXX) $(CXXFLAGS) -DBUILD_TESTS $(INCLUDES) -c $< -o $@

test: $(TEST_EXECS)
	@echo "\n=== Running Correctness Tests ==="
	@for t in $(TEST_EXECS); do echo "\n--- $$t ---"; ./$$t || exit 1; done
	@echo "\n=== All Tests Passed! ==="

performance: $(PERF_EXECS)
	@echo "\n=== Running Performance Benchmarks ==="
	@echo "\n--- $(BUILD)/performance ---"; ./$(BUILD)/performance || exit 1
	@echo "\n--- $(BUILD)/compare_bwt_performance ---"; ./$(BUILD)/compare_bwt_performance data/canterbury_corpus/alice29.txt || exit 1

clean:
	@echo "Cleaning..."
	@rm -rf $(BUILD) $(LIB)

rebuild: clean all
