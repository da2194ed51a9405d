#include "generate_main.hpp"
int main(int argc, char** argv) { return generate_main(argc, argv, 2); }
