/* oracle/api_suite_main.c -- TEST INFRASTRUCTURE: runs the reference's own encoder API unit test (src/test_libFLAC/encoders.c,
 * compiled unmodified, where it lies) against whichever library the executable is linked with (oracle/Makefile: suite). */
#include <stdio.h>
int test_encoders(void);
int main(void)
{
	const int ok = test_encoders();
	printf("\n%s\n", ok ? "ENCODER API SUITE PASSED" : "ENCODER API SUITE FAILED");
	return ok ? 0 : 1;
}
