// oracle/api_suite_main.cpp -- TEST INFRASTRUCTURE: runs the reference's own unit test of the C++ encoder API
// (src/test_libFLAC++/encoders.cpp on src/libFLAC++/stream_encoder.cpp, both compiled unmodified where they lie) against
// whichever library the executable is linked with (oracle/Makefile: suite).
#include <cstdio>
bool test_encoders();
int main()
{
	const bool ok = test_encoders();
	printf("\n%s\n", ok ? "C++ ENCODER API SUITE PASSED" : "C++ ENCODER API SUITE FAILED");
	return ok ? 0 : 1;
}
